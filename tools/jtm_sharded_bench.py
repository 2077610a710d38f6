"""JTM.optimize sharded over W workers INSIDE the library (dm_jtm_optimize_cached with a communicator) at catalogue scale.
On a one-GPU box the W worker processes share GPU 0 over the host transport (no speed-up to expect: the point is that the sharded
path runs at full size, equals the single-rank projection, and what the exchange costs); with W GPUs pass transport=rccl.
  python tools/jtm_sharded_bench.py [items=1000000] [depth=20] [world=2] [transport=host] [rows=4]"""
import faulthandler, multiprocessing as mp, os, socket, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, transport, items, depth, nrow, q):
    faulthandler.dump_traceback_later(int(os.environ.get("DM_BENCH_WATCHDOG", "600")), exit=True)
    from dismember_amd import Engine, synth
    from dismember_amd.comm import Comm
    from dismember_amd.jtm import JTM
    E, L = 128, 10
    dev = rank if transport == "rccl" else 0
    tree = synth.make_tree(items, depth, np.random.default_rng(synth.SEED))
    eng = Engine(dev)
    eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
    eng.load_weights_din_synthetic(E, (1 << (depth + 1)) - 1, synth.SEED, tree_depth=depth, rho=0.95)
    hist = synth.make_users(tree["leaf_ids"], 4 * 65536, L, np.random.default_rng(1))
    order = np.argsort(tree["leaf_ids"], kind="stable")
    pick = np.random.default_rng(2).integers(0, len(hist), size=items * nrow)
    jt = JTM.from_arrays(eng, tree["leaf_ids"][order], tree["leaf_codes"][order], depth, np.arange(items + 1, dtype=np.int64) * nrow,
                         hist[pick].reshape(-1), gap=2, seq_len=L)
    single = None
    if rank == 0:
        jt.optimize(as_array=True)                              # warm-up (scales, split copies)
        t0 = time.perf_counter(); single = jt.optimize(as_array=True); t_single = time.perf_counter() - t0
        print("rank 0 alone: %.2f s" % t_single, flush=True)
    comm = Comm(world, rank, "127.0.0.1", port, transport=transport, device_id=dev)
    jt.comm = comm
    if rank != 0:
        jt.optimize(as_array=True)                              # the other ranks warm up inside the first collective run
    else:
        jt.optimize(as_array=True)
    comm.barrier()
    tim = {}
    t0 = time.perf_counter(); proj = jt.optimize(as_array=True, timing=tim); dt = time.perf_counter() - t0
    q.put((rank, dt, zlib.crc32(proj.tobytes()), None if single is None else zlib.crc32(single.tobytes()), tim["sharding"]))
    comm.barrier()
    eng.attach_comm(None); eng.close(); comm.close()


if __name__ == "__main__":
    a = sys.argv[1:]
    items, depth, world = int(a[0]) if a else 1_000_000, int(a[1]) if len(a) > 1 else 20, int(a[2]) if len(a) > 2 else 2
    transport, nrow = a[3] if len(a) > 3 else "host", int(a[4]) if len(a) > 4 else 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, transport, items, depth, nrow, q)) for r in range(world)]
    [p.start() for p in ps]
    out = sorted((q.get(timeout=1200) for _ in range(world)), key=lambda t: t[0])
    [p.join(60) for p in ps]
    same = len({o[2] for o in out}) == 1 and out[0][2] == out[0][3]
    print("JTM.optimize sharded over %d workers (%s transport), %d items x %d rows, depth %d: %.2f s (slowest rank); projection equal on all ranks "
          "and to the single-rank run: %s" % (world, transport, items, nrow, depth, max(o[1] for o in out), same))
    for o in out:
        print("  rank %d: %.2f s  %s" % (o[0], o[1], o[4]))
    sys.exit(0 if same else 1)
