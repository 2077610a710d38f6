"""Data-parallel training check with the library's own gradient exchange (dm_train_sync_gradients =
LocalOptimizer.syncGradients, tdm/.../optim/LocalOptimizer.scala:164-187): W worker processes share ONE GPU over the
host transport (RCCL refuses two ranks per device; same entry point and kernels, only the wire differs).  After three
TDMTrainer steps (device-side negative sampling, sync, Adam) every replica must hold bit-identical weights, and they
must equal a single worker that sees all W slices with the gradient formed in rank order.
Prints one JSON line; the driver-side copy is kept as profiles/<round>_dp_train_check.json.
  usage: python tools/dp_train_check.py [workers]"""
import json, multiprocessing as mp, os, socket, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
NEG = np.arange(13, dtype=np.int32)
STEPS = 3


def data(world):
    from dismember_amd import synth
    t = np.load(os.path.join(GOLDEN, "tdm_tree.npz"))
    rng = np.random.default_rng(0)
    seqs = synth.make_users(t["leaf_ids"], 32 * world, 10, rng)
    tgt = rng.choice(t["leaf_ids"], 32 * world).astype(np.int32)
    return t, seqs, tgt


def engine(t):
    from dismember_amd import Engine
    e = Engine(0)
    e.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"])); e.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    e.load_weights_din(np.load(os.path.join(GOLDEN, "din_f32.npy")) * np.float32(0.3), 16, 8191)
    return e


def worker(rank, world, port, q):
    from dismember_amd import sharding
    from dismember_amd.comm import Comm
    from dismember_amd.trainer import TDMTrainer
    t, seqs, tgt = data(world)
    eng = engine(t)
    comm = Comm(world, rank, "127.0.0.1", port, transport="host")
    tr = TDMTrainer(eng, NEG, lr=1e-3, comm=comm, seed=77, sampler="device")
    lo, hi = sharding.shard_range(len(seqs), rank, world)
    losses = [float(tr.step(seqs[lo:hi], tgt[lo:hi])) for _ in range(STEPS)]
    q.put((rank, losses, eng.train_download("weights")))
    comm.barrier()
    eng.close(); comm.close()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    out = sorted((q.get(timeout=600) for _ in range(world)), key=lambda x: x[0])
    [p.join(120) for p in procs]
    w0 = out[0][2]
    rec = {"workers": world, "steps": STEPS, "transport": "host (W processes on one GPU)",
           "replicas_bit_identical": bool(all(np.array_equal(w0, o[2]) for o in out[1:])),
           "losses_per_worker": [o[1] for o in out],
           "weights_changed": bool(not np.array_equal(w0, np.load(os.path.join(GOLDEN, "din_f32.npy")) * np.float32(0.3))),
           "exit_codes": [p.exitcode for p in procs]}
    # single worker holding every slice: the same sampled rows (same seeds), gradients summed in rank order, one Adam step
    from dismember_amd import sharding
    t, seqs, tgt = data(world)
    ref = engine(t)
    ref.train_init(lr=1e-3)
    for it in range(STEPS):
        for r in range(world):
            lo, hi = sharding.shard_range(len(seqs), r, world)
            ref.train_step_sampled(seqs[lo:hi], tgt[lo:hi], NEG, 1, seed=77 + 1000003 * it + r, use_mask=True)   # gradients accumulate until the Adam step
        ref.adam_step(1.0 / world)
    wr = ref.train_download("weights")
    rec["max_abs_diff_vs_single_worker"] = float(np.abs(wr.astype(np.float64) - w0).max())
    rec["max_abs_weight_update"] = float(np.abs(w0.astype(np.float64) - np.load(os.path.join(GOLDEN, "din_f32.npy")) * np.float32(0.3)).max())
    rec["note_single_worker"] = ("the single worker adds worker r's rows onto the running gradient one by one, a worker sums its own rows first: "
                                 "same terms, different fp32 association, so agreement is to rounding, not bit for bit")
    ref.close()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
