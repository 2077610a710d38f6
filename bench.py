#!/usr/bin/env python3
"""bench.py — TDM beam-search serving throughput on MI355X (BASELINE.json's metric: 10M-item tree, 1M users).

Workload: synthetic 10M-item depth-24 binary tree, 128-d embeddings, DIN scorer, beam=200, topk=200, L=10
(configs/c2_tdm_serve_1m.conf).  One step = one beam search over ONE shard of 131072 users whose
histories are already resident in HBM; the steps cycle through 8 distinct shards (the 1M-user population).
N GPUs = N independent user populations (replicated table, no data-path collective; the harness barrier and the
slowest-rank clock use torch.distributed, the training extra's gradient exchange dismember_amd.comm = RCCL in the library).

Prints ONE compact JSON line (rank 0, < 4 KB: compact_line()); the full result object goes to bench_full.json (write_full()).
`--gpus N` with no launcher around it starts its own N ranks (launch_ranks()).
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_MFMA_F16_TFLOPS = 2516.6  # v_mfma_f32_16x16x32_f16: 16384 FLOP / 16 cycles / SIMD x 1024 SIMDs x 2.4 GHz (the guide's "~2.5 PF dense")


def roofline(mode, rows, avg_ms, E, L, kernel=None):
    """MFMA roofline of the beam kernel for one scorer arithmetic.  achieved = ALGORITHMIC flops of this formulation
    (2(E^2 + 2LE + E) per scored row) / the kernel's average launch duration (HIP events on its stream).
    peak: fp32-input MFMA peak for the f32 mode; for the split mode every product costs 3 fp16 MFMAs on padded 16x16x32
    tiles, so the pipe's dense fp16 peak is scaled by algorithmic / issued flops (the time the matrix pipe minimally
    needs per row is what bounds the kernel).  Beside `frac` the split mode reports the plain fractions of the dense fp16
    peak: issued (what the matrix pipe executes), useful (this formulation's flops) and the reference formulation's
    (2(2LE + 3E^2 + E) per row — the reference recomputes the history projection per candidate)."""
    kq = (L + 3) // 4
    nt = E // 16
    flops_own = 2 * (E * E + 2 * L * E + E)
    flops_ref = 2 * (2 * L * E + 3 * E * E + E)
    t = avg_ms * 1e-3
    ach = rows * flops_own / t / 1e12
    if mode == "f32":
        issued = (nt * 4 + kq * nt + nt * 4 * nt) * 2048 / 16.0            # S^T + P x G (ceil(L/4) k-steps) + main chain, 16x16x4
        peak = PEAK_MFMA_F32_TFLOPS
        extra = {"mfma_issued_tflops": rows * issued / t / 1e12}
    else:
        ns = E // 32
        issued = (3 * ns + 2 * nt + 3 * ns * nt) * 16384 / 16.0              # 3 x S^T + 2 x (P x G) + 3 x main chain, 16x16x32
        peak = PEAK_MFMA_F16_TFLOPS * flops_own / issued
        iss_t = rows * issued / t / 1e12
        extra = {"mfma_issued_tflops_f16": iss_t, "f16_dense_peak_tflops": PEAK_MFMA_F16_TFLOPS,
                 "issued_frac_of_fp16_peak": iss_t / PEAK_MFMA_F16_TFLOPS,
                 "useful_frac_of_fp16_peak": ach / PEAK_MFMA_F16_TFLOPS,
                 "issued_per_algorithmic_flop": issued / flops_own,
                 "peak_note": "fp16 dense peak x algorithmic/issued flops: 3 fp16 MFMAs per product (hi*hi + hi*lo + lo*hi; 2 for the attention-combine product) on 16x16x32 tiles"}
    r = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
         "kernel": kernel or ("dm_beam_kernel<%d, %d, %s>" % (E, kq, "true" if mode != "f32" else "false")), "kernel_ms_avg": avg_ms,
         "flops_per_row_algorithmic": flops_own,
         "reference_formulation_flops_per_row": flops_ref,
         "reference_formulation_tflops": rows * flops_ref / t / 1e12}
    r.update(extra)
    return r
PEAK_HBM_GBPS = 8000.0


_T0 = time.perf_counter()


class ClockSampler:
    """Shader clock and socket power of GPU `dev` while the timed steps run (rocm-smi every ~0.1 s from a host thread; the headline
    kernel runs power-managed: DESIGN.md §4).  median() -> (sclk MHz, watts) or (None, None) when rocm-smi is not usable."""

    def __init__(self, dev):
        import threading
        self.dev, self.clk, self.pw, self._stop = int(dev), [], [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
                lines = [ln for ln in out.splitlines() if ln.strip()]
                hdr, row = lines[0].split(","), lines[1].split(",")
                for k_, v_ in zip(hdr, row):
                    if k_.strip().lower().startswith("sclk clock speed"):
                        m_ = re.search(r"(\d+)\s*mhz", v_.lower())
                        if m_:
                            self.clk.append(int(m_.group(1)))
                    if "power" in k_.lower() and "(w)" in k_.lower():
                        try:
                            self.pw.append(float(v_))
                        except ValueError:
                            pass
            except Exception:      # noqa: BLE001
                return
            self._stop.wait(0.1)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join(6)

    def median(self):
        busy = [c for c in self.clk if c > 500]           # (samples taken before / after the kernels sit at the idle clock)
        return (float(np.median(busy)) if busy else None, float(np.median(self.pw)) if self.pw else None)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--users", type=int, default=131072, help="users per step (= per shard) per GPU")
    ap.add_argument("--shards", type=int, default=8, help="distinct user shards the steps cycle through (8 x 131072 = the 1M-user population of the target)")
    ap.add_argument("--items", type=int, default=10_000_000)
    ap.add_argument("--depth", type=int, default=24)
    ap.add_argument("--conf", default=os.path.join(ROOT, "configs", "c2_tdm_serve_1m.conf"),
                    help="TDMTrainDeepModel .conf the serving / training parameters come from (reference format, dismember_amd/conf.py)")
    ap.add_argument("--dr-conf", default=os.path.join(ROOT, "configs", "c5_dr_10m.conf"), help="DeepRetrieval .conf of the Deep-Retrieval extra")
    ap.add_argument("--embed", type=int, default=None, help="override model.embed_size")
    ap.add_argument("--beam", type=int, default=None, help="override model.beam_size")
    ap.add_argument("--topk", type=int, default=None, help="override model.topk_number")
    ap.add_argument("--seq-len", type=int, default=None, help="override model.seq_len")
    ap.add_argument("--cpu-users", type=int, default=-1, help="oracle sample size (-1 auto, 0 skip)")
    ap.add_argument("--rho", type=float, default=0.95, help="parent-child correlation of the synthetic node embeddings")
    ap.add_argument("--train", type=int, default=1, help="also time a training step (0 = skip)")
    ap.add_argument("--small", type=int, default=1, help="also time BASELINE configs[1] (1M-item depth-20 tree) and the training step on it; 0 = skip")
    ap.add_argument("--big", type=int, default=None, help=argparse.SUPPRESS)   # old name of --small
    ap.add_argument("--dr", type=int, default=1, help="also time Deep-Retrieval serving (config 5: D=3, K=1000, beam=50, 10M items); 0 = skip")
    ap.add_argument("--scorer", default="auto", choices=["auto", "f32", "split_f16"],
                    help="arithmetic of the headline run (include/dismember_hip.h: dm_set_scorer_mode; auto = the library default)")
    ap.add_argument("--other-scorer", type=int, default=1, help="also time the OTHER scorer arithmetic on the same engine and inputs (0 = skip)")
    ap.add_argument("--recall-users", type=int, default=1024, help="users for recall@topk vs brute force (0 skip)")
    ap.add_argument("--diverse", type=int, default=1, help="also time the headline search on beams that DIVERGE (attention path x1.7, embeddings x32) and on an iid (rho 0) table; 0 = skip")
    ap.add_argument("--long-history", type=int, default=1, help="also time the headline search with 24-position histories (fused two-key-tile kernel); 0 = skip")
    ap.add_argument("--host-buffer-steps", type=int, default=3, help="steps of the headline workload through the host-buffer entry point (0 = skip)")
    ap.add_argument("--jtm-full", type=int, default=1, help="also time the FULL JTM.optimize over the 10M-item catalogue (BASELINE configs[3]); 0 = skip")
    ap.add_argument("--jtm-rows", type=int, default=4, help="training rows per item of the full JTM.optimize extra")
    ap.add_argument("--trained-recall-steps", type=int, default=6000, help="TDMTrainer steps (1024 tree-consistent targets each) before recall@topk vs brute force is measured again on the 1M-item tree; 0 = skip")
    ap.add_argument("--otm64", type=int, default=1, help="also time OTM serving and one OTM training iteration in the reference's fp64 (BASELINE configs[2]); 0 = skip")
    a = ap.parse_args()
    if a.big is not None:
        a.small = a.big
    from dismember_amd import conf as dmconf
    a.params = dmconf.task_params("TDMTrainDeepModel", a.conf)
    a.embed = a.embed or a.params["embed_size"]
    a.beam = a.beam or a.params["beam_size"]
    a.topk = a.topk or a.params["topk_number"]
    a.seq_len = a.seq_len or a.params["seq_len"]
    a.dr_params = dmconf.read_conf(a.dr_conf, "model")
    return a


def effective_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 logical CPUs but runs the container under a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def host_mem_available():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
                break
        else:
            return 0
        try:
            lim = open("/sys/fs/cgroup/memory.max").read().strip()
            cur = int(open("/sys/fs/cgroup/memory.current").read().strip())
            if lim != "max":
                avail = min(avail, int(lim) - cur)
        except Exception:
            pass
        return avail
    except Exception:
        return 0


def cpu_baseline(tree, w, E, L, num_index, seqs, beam, topk, n_users):
    """The CPU oracle (restatement of the reference's Scala path, oracle/) timed on the host cores,
    one worker thread per core over contiguous user ranges like T/evaluation/Evaluator.scala:28-37."""
    from oracle import pyoracle as po
    po.build()
    otree = po.TdmTree(tree["codes"], tree["ids"], tree["is_leaf"], tree["leaf_ids"], tree["leaf_codes"],
                       tree["max_level"])
    din = po.Din(w, E, L, num_index)
    cores = effective_cores()
    t0 = time.perf_counter()
    otree.recommend_batch(din, seqs[:2], topk, beam, n_threads=1)
    per_user = (time.perf_counter() - t0) / 2
    # one core first: ~6 s of the same users on ONE thread (the reference's per-thread rate, T/evaluation/Evaluator.scala runs one
    # such loop per core)
    n1 = int(max(4, min(seqs.shape[0], 6.0 / max(per_user, 1e-6))))
    t0 = time.perf_counter()
    otree.recommend_batch(din, seqs[:n1], topk, beam, n_threads=1)
    dt1 = time.perf_counter() - t0
    one = dict(value=n1 / dt1, unit="users/s", cores=1, kind="port",
               sample="the first %d users of the same workload on one thread, %.1f s wall, oracle/libdm_oracle.so" % (n1, dt1))
    if n_users <= 0:                                   # auto: aim at ~12 s of wall time on all cores
        n_users = int(max(cores, min(16384, 12.0 * cores / max(per_user, 1e-6))))
    n_users = min(n_users, seqs.shape[0])
    cores = min(cores, n_users)
    t0 = time.perf_counter()
    ids, sc, cnt = otree.recommend_batch(din, seqs[:n_users], topk, beam, n_threads=cores)
    dt = time.perf_counter() - t0
    return dict(value=n_users / dt, unit="users/s", cores=cores, kind="port", sample_short="%d users of the headline workload, %.1f s, %d threads" % (n_users, dt, cores),
                sample="%d users of the same workload, %.1f s wall, oracle/libdm_oracle.so (C restatement of the "
                       "reference path, one pthread per usable core (cgroup quota) over contiguous user ranges)" % (n_users, dt)), one, (ids, cnt, otree, din)


def near_tie_report(eng, otree, din, seqs_diff, beam, topk, modes=None, max_users=4096):
    """"Every differing user is a near-tie at a cut", measured (tests/helpers.py: explain_users).  For users whose end-to-end id lists
    differ — device vs CPU oracle (modes=None), or the two device arithmetics (modes=(a, b)) — trace both searches, find the first
    prune whose ordered outcome differs and compare the score gaps of the candidates that changed order with the stated tolerance
    (atol 1e-5 + rtol 1e-4 |s| per score, so 2x that between two candidates).  Checker code: the oracle's integer logic replays the cuts."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import explain_users, trace_levels
    n = min(len(seqs_diff), max_users)
    if n == 0:
        return {"differing_users_analysed": 0, "explained_by_near_tie": 0, "max_cut_gap": 0.0}
    sq = np.ascontiguousarray(seqs_diff[:n])
    if modes is None:
        _, _, _, tc, ts, tn = eng.tdm_beam_search_trace(sq, beam, topk)
        ta = [trace_levels(tc, ts, tn, u) for u in range(n)]
        tb = [otree.recommend(din, sq[u], topk, beam, trace=True)[2] for u in range(n)]
    else:
        keep = eng.scorer_mode()["setting"] if "setting" in eng.scorer_mode() else None
        tr = []
        for m in modes:
            eng.set_scorer_mode(m)
            _, _, _, tc, ts, tn = eng.tdm_beam_search_trace(sq, beam, topk)
            tr.append([trace_levels(tc, ts, tn, u) for u in range(n)])
        eng.set_scorer_mode(keep or "auto")
        ta, tb = tr
    r = explain_users(otree, beam, topk, ta, tb)
    out = {"differing_users_analysed": n, "users_with_a_diverging_cut": r["differing_users"], "explained_by_near_tie": r["explained_by_near_tie"],
           "max_cut_gap": r["max_cut_gap"], "max_cut_gap_over_allowance": r["max_cut_gap_over_tol"], "first_diverging_level_histogram": r["by_level"],
           "allowance": "2 x (1e-5 + 1e-4 |s|) between two candidates the sides order differently, on both sides' scores"}
    if r["unexplained"]:
        out["unexplained"] = [{"user": int(u), **{k: (v if not isinstance(v, float) or np.isfinite(v) else None) for k, v in d.items()}} for u, d in r["unexplained"][:4]]
    return out


def init_distributed():
    """The launcher contract (torchrun env).  torch.distributed carries the HARNESS: the barrier and the slowest-rank clock the
    bench contract asks for.  What the product exchanges (the training extra's gradients) runs over dismember_amd.comm — RCCL
    inside the library — created by make_comm() right before it is needed, so that a rendezvous problem there can never cost
    the headline line."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return None, rank, world, local
    import torch
    import torch.distributed as dist
    if os.environ.get("DM_BENCH_BACKEND", "gloo" if "DM_FORCE_DEVICE" in os.environ else "nccl") == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:                                         # single-GPU smoke test of the N > 1 path (DM_FORCE_DEVICE=0)
        dist.init_process_group("gloo")
    return dist, rank, world, local


def make_comm(dist, rank, world, local):
    """The library's communicator for the ranks of this job: rank 0 picks a free port and the harness broadcasts it.  RCCL
    (ncclCommInitRank over the job's GPUs) by default — and ONLY RCCL: a rendezvous failure is reported in the line (`comm_error`, the
    extras that need the communicator say why they did not run) instead of quietly falling back to the host transport, whose numbers
    would read like a scaling result.  DM_COMM_TRANSPORT=host asks for the host TCP transport explicitly (several ranks on one GPU)."""
    import socket
    from dismember_amd.comm import Comm
    dev = int(os.environ.get("DM_FORCE_DEVICE", local))
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    transport = os.environ.get("DM_COMM_TRANSPORT", "rccl")
    port = [0]
    if rank == 0:
        s_ = socket.socket(); s_.bind(("", 0)); port[0] = s_.getsockname()[1]; s_.close()
    dist.broadcast_object_list(port, src=0)
    err = None
    try:
        c = Comm(world, rank, addr, port[0], transport, dev)
    except Exception as ex:       # noqa: BLE001 — any failure here only costs the extras that exchange data
        c, err = None, repr(ex)
    errs = [None] * world
    dist.all_gather_object(errs, err)
    bad = [(r_, e_) for r_, e_ in enumerate(errs) if e_]
    if not bad:
        return c, transport, None
    if c is not None:
        c.close()
    return None, None, ("the %s communicator could not be created on rank %d: %s (no silent fallback: set DM_COMM_TRANSPORT=host to run the "
                        "exchange over the host transport)" % (transport, bad[0][0], bad[0][1]))


def stage(msg):
    """Progress on stderr (DM_BENCH_VERBOSE=1): the JSON line on stdout stays the only output the driver parses."""
    if os.environ.get("DM_BENCH_VERBOSE"):
        sys.stderr.write("[bench rank %s %.1fs] %s\n" % (os.environ.get("RANK", "0"), time.perf_counter() - _T0, msg)); sys.stderr.flush()


def sustained_mfma_tflops():
    """Mean sustained rates of tools/mfma_power_microbench.hip from its committed log: {'f16_quiet', 'f16_random', 'f64_quiet', 'f64_random'} (TFLOP/s)."""
    out, acc = {}, {}
    try:
        for line in open(os.path.join(ROOT, "profiles", "r05_mfma_power.log")):
            if "TFLOP/s" not in line:
                continue
            if "f64_16x16x4" not in line and "16x16x32_f16" not in line:      # (the log also carries the 32x32x16 shape: 1.56 PFLOP/s on random operands)
                continue
            key = ("f64" if "f64_16x16x4" in line else "f16") + ("_random" if "random" in line else "_quiet")
            acc.setdefault(key, []).append(float(line.split("TFLOP/s")[0].split()[-1]))
        out = {k: sum(v) / len(v) for k, v in acc.items()}
    except Exception:      # noqa: BLE001 — the log is an annotation, not part of the measurement
        pass
    return out


def _r(x, nd=4):
    """Round floats for the compact line (keeps it short; the full object keeps every digit)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    if abs(x) >= 1000:
        return round(x, 1)
    return float("%.*g" % (nd + 1, x))


def _dig(o, *path):
    for k in path:
        if not isinstance(o, dict) or k not in o:
            return None
        o = o[k]
    return o


COMPACT_LIMIT = 4096
COMPACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline")


def compact_line(full, full_path=None):
    """The ONE stdout line the driver parses: the contract's keys, `roofline`, `cpu_baseline` and one scalar per extra, no prose,
    under COMPACT_LIMIT bytes (tests/test_bench_line.py).  Everything else lives in the full object (bench_full.json)."""
    cfg, roof, cpu = full.get("config", {}), full.get("roofline", {}), full.get("cpu_baseline")
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    out["value"], out["ms_per_step"] = _r(out["value"], 6), _r(out["ms_per_step"], 5)
    out["dtype"] = "f32"            # the path computes in fp32 (the split scorer feeds fp32 values as fp16 hi+lo pairs, fp32 accumulation)
    out["data"] = "synthetic"
    out["config"] = {"workload": str(cfg.get("workload", ""))[:160]}
    for k in ("items", "depth", "embed", "beam", "topk", "seq_len", "users_per_step_per_gpu", "user_shards", "parallelism", "scored_rows_per_user", "scorer"):
        if k in cfg:
            out["config"][k] = _r(cfg[k], 5)
    out["roofline"] = {k: _r(roof.get(k), 5) for k in ("bound", "kernel", "kernel_ms_avg", "achieved", "peak", "unit", "frac", "traffic")}
    for k_out, k_in in (("useful_frac_of_fp16_peak", "useful_frac_of_fp16_peak"), ("issued_frac_of_fp16_peak", "issued_frac_of_fp16_peak"),
                        ("issued_frac_of_sustained", "issued_frac_of_sustained_random_operand_rate"), ("launches", "launches"),
                        ("clock_mhz", "clock_mhz_under_load"), ("power_w", "socket_power_w_under_load")):
        if roof.get(k_in) is not None:
            out["roofline"][k_out] = _r(roof[k_in])
    if isinstance(cpu, dict):
        out["cpu_baseline"] = {"value": _r(cpu.get("value"), 5), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                               "sample": str(cpu.get("sample_short") or cpu.get("sample", ""))[:96]}
    else:
        out["cpu_baseline"] = None
    rec = full.get("recall_at_%s_vs_bruteforce" % cfg.get("topk", 200)) or full.get("recall_at_200_vs_bruteforce")
    out["recall_at_200_vs_bruteforce"] = _r(rec.get("value") if isinstance(rec, dict) else rec)
    scal = {
        "recall_at_200_vs_bruteforce_trained": _dig(full, "extra_trained_recall", "recall_trained"),
        "recall_at_200_vs_bruteforce_before_training": _dig(full, "extra_trained_recall", "recall_untrained"),
        "target_in_top200_trained": _dig(full, "extra_trained_recall", "target_in_beam_topk_trained"),
        "id_lists_identical_to_cpu_oracle": full.get("id_lists_identical_to_cpu_oracle"),
        "id_lists_compared_with_cpu_oracle": full.get("id_lists_compared_with_cpu_oracle"),
        "near_tie_explained_frac": full.get("near_tie_explained_frac"),
        "cpu_baseline_1core_users_per_s": _dig(full, "cpu_baseline_1core", "value"),
        "host_buffer_users_per_s": full.get("host_buffer_users_per_s"),
        "rccl_nranks": full.get("rccl_nranks"),
        "comm_transport": full.get("comm_transport"),
        "per_rank_users_per_s": full.get("per_rank_users_per_s"),
        "diverse_beams_users_per_s": full.get("diverse_beams_users_per_s"),
        "iid_table_users_per_s": _dig(full, "extra_diverse_beams", "iid_table_rho0", "users_per_s"),
        "other_scorer_f32_users_per_s": _dig(full, "extra_other_scorer", "users_per_s"),
        "other_scorer_f32_frac": _dig(full, "extra_other_scorer", "roofline", "frac"),
        "long_history_L24_users_per_s": full.get("long_history_L24_users_per_s"),
        "otm_serve_users_per_s": _dig(full, "extra_otm_serve", "users_per_s"),
        "otm_fp64_users_per_s": _dig(full, "extra_otm_fp64", "users_per_s"),
        "otm_fp64_frac": _dig(full, "extra_otm_fp64", "roofline", "frac"),
        "otm_fp64_train_iter_8192_s": _dig(full, "extra_otm_fp64", "train_iteration_batch_8192", "seconds"),
        "jtm_optimize_s": _dig(full, "extra_jtm_optimize", "seconds"),
        "jtm_optimize_din_rows_per_s": _dig(full, "extra_jtm_optimize", "din_rows_per_s"),
        "jtm_rows_kernel_gather_frac": _dig(full, "extra_jtm_optimize", "roofline", "frac"),
        "jtm_rows_kernel_issued_frac_fp16": _dig(full, "extra_jtm_optimize", "roofline", "issued_frac_of_fp16_peak"),
        "jtm_rows_kernel_ms_avg": _dig(full, "extra_jtm_optimize", "roofline", "kernel_ms_avg"),
        "jtm_scoring_items_per_s": _dig(full, "extra_jtm_scoring", "items_per_s"),
        "c1_1m_tree_users_per_s": _dig(full, "extra_1m_item_tree", "users_per_s"),
        "c1_1m_tree_frac": _dig(full, "extra_1m_item_tree", "roofline", "frac"),
        "c1_1m_tree_recall_at_200": _dig(full, "extra_1m_item_tree", "recall_at_200_vs_bruteforce"),
        "train_step_ms": _dig(full, "extra_train_step", "ms_per_step"),
        "dr_f64_users_per_s": _dig(full, "extra_deep_retrieval", "beam_search_users_per_s"),
        "dr_f64_kernel_ms": _dig(full, "extra_deep_retrieval", "beam_search_kernel_ms_per_step"),
        "dr_f32_users_per_s": _dig(full, "extra_deep_retrieval", "f32_split", "beam_search_users_per_s"),
        "dr_f32_kernel_ms": _dig(full, "extra_deep_retrieval", "f32_split", "beam_search_kernel_ms_per_step"),
        "c0_tdm_recommend_ms": _dig(full, "extra_config0_latency", "ms_per_call"),
        "c0_cpu_oracle_ms": _dig(full, "extra_config0_latency", "cpu_oracle_ms_per_call"),
        "c0_recall_trained_beam200": _dig(full, "extra_config0_latency", "recall_vs_bruteforce_trained_model", "beam200_top200"),
        "comm_error": (str(full["comm_error"])[:120] if full.get("comm_error") else None),
    }
    for k, v in scal.items():
        if v is None:
            continue
        out[k] = [_r(x, 5) for x in v] if isinstance(v, (list, tuple)) else _r(v, 5)
    if full_path:
        out["full"] = full_path
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:                    # cannot happen with the keys above; never let an extra cost the headline
        for k in list(scal):
            out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def write_full(full):
    """The whole result object (the lab notebook) goes to files, not to stdout: bench_full.json beside bench.py and, when the directory
    exists, gpurun_out/bench_full.json.  Returns the name quoted in the compact line."""
    txt = json.dumps(full, indent=1)
    name = None
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    f.write(txt + "\n")
                name = name or os.path.relpath(os.path.join(d, "bench_full.json"), ROOT)
        except OSError:
            pass
    return name


def launch_ranks(a):
    """`python bench.py --gpus N` as typed (no torchrun around it): start N ranks of this same command, one per GPU, with the launcher
    environment init_distributed() reads; rank 0 prints the line.  Fails loudly when the node has fewer than N devices (unless
    DM_FORCE_DEVICE pins every rank to one device for a smoke test of the N > 1 path)."""
    import socket
    import subprocess
    from dismember_amd import _native
    n = C.c_int(0)
    rc = _native.lib().dm_device_count(C.byref(n))
    if rc != 0:
        sys.exit("bench.py: dm_device_count failed (%d): no usable HIP device" % rc)
    if "DM_FORCE_DEVICE" not in os.environ and n.value < a.gpus:
        sys.exit("bench.py: --gpus %d asked for, %d HIP device(s) visible; refusing to run a smaller job under that label "
                 "(DM_FORCE_DEVICE=0 DM_COMM_TRANSPORT=host runs every rank on one device as a smoke test)" % (a.gpus, n.value))
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [None] * a.gpus
    try:
        while any(c is None for c in rcs):
            for i, p_ in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = p_.poll()
            if any(c not in (None, 0) for c in rcs):
                break
            time.sleep(0.2)
    finally:
        for i, p_ in enumerate(procs):             # a failed rank takes the job down: its peers would wait in a barrier for ever
            if p_.poll() is None:
                p_.terminate()
        for p_ in procs:
            try:
                p_.wait(20)
            except subprocess.TimeoutExpired:
                p_.kill()
    bad = [(i, c) for i, c in enumerate(rcs) if c not in (0,)]
    if bad:
        sys.exit("bench.py: rank %d exited with %s" % (bad[0][0], bad[0][1]))
    sys.exit(0)



def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (a.gpus, os.environ.get("WORLD_SIZE", "1")))
    wd = int(os.environ.get("DM_BENCH_WATCHDOG", "0"))
    if wd > 0:                         # debugging aid: dump every thread's stack and exit if the run is still alive after `wd` seconds
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    from dismember_amd import sharding
    dist, rank, world, local = init_distributed()
    comm = comm_transport = comm_err = None
    jtm_comm_err_top = None
    torch = None
    if dist is not None:
        import torch
    from dismember_amd import Engine
    from dismember_amd import synth

    E, L, depth = a.embed, a.seq_len, a.depth
    num_index = (1 << (depth + 1)) - 1
    rng = np.random.default_rng(synth.SEED)
    tree = synth.make_tree(a.items, depth, rng)
    # every rank: its own population of `shards` x `users` users (rank-dependent seeds), one device buffer per shard
    NSH = max(1, a.shards)
    shard_seqs = [synth.make_users(tree["leaf_ids"], a.users, L, np.random.default_rng(synth.SEED + 1 + rank + 1000 * k)) for k in range(NSH)]
    seqs = shard_seqs[0]

    eng = Engine(int(os.environ.get("DM_FORCE_DEVICE", local)))   # override only for single-GPU smoke tests of the N>1 path
    eng.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth)
    eng.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
    # table generated on the device (same bits on every rank); tree-correlated rows so that the beam has
    # something to follow (iid rows make recall vs brute force ~0 by construction)
    eng.load_weights_din_synthetic(E, num_index, synth.SEED, tree_depth=depth, rho=a.rho)

    U = a.users
    d_seqs = []
    for k in range(NSH):
        d_ = eng.dev_alloc(U * L * 4)
        eng.h2d(d_, shard_seqs[k])
        d_seqs.append(d_)
    d_seq = d_seqs[0]
    d_ids = eng.dev_alloc(U * a.topk * 4)
    d_sc = eng.dev_alloc(U * a.topk * 4)
    d_cnt = eng.dev_alloc(U * 4)

    def sync():
        eng.synchronize()
        if torch is not None:
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t_ = torch.tensor([x], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        return float(t_.item())

    eng.set_scorer_mode(a.scorer)
    # untimed: one pass per shard records its scored-row count (the roofline's work figure) and builds the scorer's planes
    shard_rows = []
    for k in range(NSH):
        eng.tdm_beam_search_dev(d_seqs[k], U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)
        sync()
        shard_rows.append(eng.last_scored_rows())
    for i in range(a.warmup):
        eng.tdm_beam_search_dev(d_seqs[i % NSH], U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)
    sync()
    eng.timing_reset()
    barrier(); sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        eng.tdm_beam_search_dev(d_seqs[i % NSH], U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)
    sync(); barrier()
    dt_own = time.perf_counter() - t0
    dt = max_over_ranks(dt_own)
    per_rank_rate = [U * a.steps / dt_own]
    if dist is not None:
        per_rank_rate = [None] * world
        dist.all_gather_object(per_rank_rate, U * a.steps / dt_own)
    n_launch, kernel_ms = eng.timing_get_kind(0)                # the search kernel proper (HIP events on the library's stream)
    n_defer, defer_ms = eng.timing_get_kind(1)                  # second pass over deferred users (one-wave kernel only)
    kern_name = eng.last_beam_kernel()
    # the clock the kernel actually runs at (it holds the package at its power limit): sampled over a SECOND, untimed run of the same steps
    # so that the sampler's host thread cannot touch the timed region
    clk_mhz = pw_w = None
    if rank == 0:
        with ClockSampler(int(os.environ.get("DM_FORCE_DEVICE", local))) as cs_:
            for i in range(max(a.steps, 12)):
                eng.tdm_beam_search_dev(d_seqs[i % NSH], U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)
            eng.synchronize()
        clk_mhz, pw_w = cs_.median()
    barrier()
    rows = sum(shard_rows[i % NSH] for i in range(a.steps)) / float(a.steps)     # scored (node, user) rows per step, averaged over the timed steps

    # results of shard 0 (recall, the CPU oracle's comparison and the other-scorer comparison all use shard 0)
    eng.tdm_beam_search_dev(d_seq, U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)
    sync()
    ids = np.empty((U, a.topk), np.int32)
    sc = np.empty((U, a.topk), np.float32)
    cnt = np.empty(U, np.int32)
    eng.d2h(ids, d_ids); eng.d2h(sc, d_sc); eng.d2h(cnt, d_cnt)

    # the same step through the HOST-buffer entry point (dm_tdm_beam_search: the request goes up and ids / scores / counts come down
    # over PCIe inside the call) — reported beside `value`, never as `value`
    host_rate = None
    if a.host_buffer_steps > 0:
        hout = (np.empty((U, a.topk), np.int32), np.empty((U, a.topk), np.float32), np.empty(U, np.int32))     # a serving loop's own result buffers
        eng.tdm_beam_search(shard_seqs[0], a.beam, a.topk, out=hout)             # shard 0: the device-resident results downloaded above
        same_as_dev = bool(np.array_equal(hout[0], ids) and np.array_equal(hout[2], cnt) and np.array_equal(hout[1], sc))
        sync(); barrier()
        t0 = time.perf_counter()
        for i in range(a.host_buffer_steps):
            eng.tdm_beam_search(shard_seqs[i % NSH], a.beam, a.topk, out=hout)
        sync(); barrier()
        dth = max_over_ranks(time.perf_counter() - t0)
        host_rate = {"host_buffer_users_per_s": world * U * a.host_buffer_steps / dth, "steps": a.host_buffer_steps,
                     "ms_per_step": dth / a.host_buffer_steps * 1e3, "identical_to_device_resident_results": same_as_dev,
                     "what": "the same users through dm_tdm_beam_search (host numpy buffers in and out: %d B up and %d B down per user over PCIe, "
                             "pageable memory; the request is cut into chunks of users whose downloads run under the kernels of the chunks "
                             "behind them, DM_HOST_PIPELINE)" % (4 * L, 8 * a.topk + 4)}

    stage("extra the same search with the OTHER scorer arithmetic (same engine, s")
    # ---- extra: the same search with the OTHER scorer arithmetic (same engine, same users) ----
    mode = eng.scorer_mode()["mode"]            # arithmetic in effect for the headline run
    info = eng.scorer_mode()
    other = None
    if a.other_scorer and E % 32 == 0:
        omode = "f32" if mode == "split_f16" else "split_f16"
        eng.set_scorer_mode(omode)
        eng.tdm_beam_search_dev(d_seq, U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)     # warms up (split: builds the fp16 planes)
        sync(); eng.timing_reset(); barrier()
        t0 = time.perf_counter()
        n_o = max(2, a.steps // 2)
        for _ in range(n_o):
            eng.tdm_beam_search_dev(d_seq, U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)
        sync(); barrier()
        dto_ = max_over_ranks(time.perf_counter() - t0)
        nlo_, kmo_ = eng.timing_get_kind(0)
        kern_o = eng.last_beam_kernel()
        ids_o = np.empty((U, a.topk), np.int32); sc_o = np.empty((U, a.topk), np.float32); cnt_o = np.empty(U, np.int32)
        eng.d2h(ids_o, d_ids); eng.d2h(sc_o, d_sc); eng.d2h(cnt_o, d_cnt)
        if omode == "split_f16":
            info = eng.scorer_mode()
        eng.set_scorer_mode(a.scorer)
        same_rows = (ids_o == ids).all(axis=1) & (cnt_o == cnt)
        dsc = np.abs(sc_o[same_rows].astype(np.float64) - sc[same_rows])
        other = {"mode": omode, "what": "the same search (engine, shard 0 of the users, beam logic) with the other scorer arithmetic "
                                        "(include/dismember_hip.h: dm_set_scorer_mode)",
                 "users_per_s": world * U * n_o / dto_, "ms_per_step": dto_ / n_o * 1e3, "steps": n_o,
                 "roofline": roofline(omode, shard_rows[0], kmo_ / max(nlo_, 1), E, L, kern_o),
                 "headline_speedup_over_this": (dto_ / n_o) / (dt / a.steps),
                 "identical_id_lists_vs_headline": "%d/%d" % (int(same_rows.sum()), U),
                 "max_abs_score_diff_on_identical_lists": float(dsc.max()) if dsc.size else None,
                 "max_abs_score": float(np.abs(sc).max())}

    stage("extra histories of 24 positions (the fused two-key-tile kernel)")
    # ---- extra: the headline search with histories of 24 positions (round-4 verdict, next #8): seq_len 17 .. 32 runs inside the fused LDS-fed
    # kernel through a second 16-position key tile (dm_beam_kernel<E, 4, SPLIT, 2>); DM_LONG_PIPELINE=1 = round 4's per-level pipeline.
    long_hist = None
    if a.long_history and rank == 0:
        L2_, U2_ = 24, min(U, 32768)
        seqs24 = synth.make_users(tree["leaf_ids"], U2_, L2_, np.random.default_rng(synth.SEED + 77))
        d_s24 = eng.dev_alloc(U2_ * L2_ * 4)
        eng.h2d(d_s24, seqs24)
        long_hist = {"seq_len": L2_, "users_per_step": U2_, "what": "the headline tree, table and beam with 24-position histories, device-resident request"}
        # (the per-level pipelines of tdm_pipeline.hip.inc / otm64.hip.inc are a TEST cross-check since round 6 — DM_LONG_PIPELINE=1, tests/test_gpu_edges.py —
        #  never selected by dispatch and no longer timed here)
        os.environ.pop("DM_LONG_PIPELINE", None)
        eng.tdm_beam_search_dev(d_s24, U2_, L2_, a.beam, a.topk, d_ids, d_sc, d_cnt)
        sync()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.tdm_beam_search_dev(d_s24, U2_, L2_, a.beam, a.topk, d_ids, d_sc, d_cnt)
        sync()
        dt24 = (time.perf_counter() - t0) / 3
        long_hist["fused"] = {"kernel": eng.last_beam_kernel(), "ms_per_step": dt24 * 1e3, "users_per_s": U2_ / dt24, "scored_rows": eng.last_scored_rows()}
        long_hist["fused_over_headline_rate"] = long_hist["fused"]["users_per_s"] / (U * a.steps / dt)      # (per GPU, L = 24 against the headline L)
        eng.dev_free(d_s24)
        eng.tdm_beam_search_dev(d_seq, U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)      # (restore shard 0's results in the output buffers)
        sync()
    barrier()

    stage("extra the headline search on beams that diverge / on an iid table")
    # ---- extra: the same search where the users' beams DIVERGE (round-4 verdict, next #2).  With the reference's init the history term of the
    # logit is several times smaller than the node term: the beams of different users overlap almost completely (distinct candidate
    # rows per level: 0.1-0.2 % of users x candidates) and 97 % of the gathers hit L2.  Scaling the attention path (att.W, W1b) by 1.7
    # and the embeddings by 32 (a sharp softmax) is the strongest history dependence that still fits the split scorer's fp16 range for
    # every user (tools/diverse_bench.py explores the neighbourhood; beyond it users fall back to the LDS-fed kernel's fp32 product);
    # it makes the beams user-specific (5-45 % distinct rows on the lower levels, 300 x more distinct result items).  Beside it: the
    # literal SURVEY §8d table (iid rows, rho 0) with the reference init.  Same tree, same users, device-resident request.
    diverse = None
    if a.diverse and (a.items, a.depth) == (10_000_000, 24) and mode != "f32":
        def _variant(att_scale, emb_scale, rho_):
            r_ = np.random.default_rng(int(synth.SEED))
            small_ = np.zeros(3 * E * E + 2 * E + 1, np.float32)
            small_[:3 * E * E] = r_.standard_normal(3 * E * E, dtype=np.float32) * 0.05
            small_[3 * E * E + E:3 * E * E + 2 * E] = r_.standard_normal(E, dtype=np.float32) * 0.05
            small_[:E * E] *= att_scale
            small_[E * E:3 * E * E].reshape(E, 2 * E)[:, E:] *= att_scale
            e_ = Engine(int(os.environ.get("DM_FORCE_DEVICE", local)))
            try:
                e_.load_tree(tree["codes"], tree["ids"], tree["is_leaf"], depth); e_.load_id_maps(tree["leaf_ids"], tree["leaf_codes"])
                e_.load_weights_din_synthetic(E, num_index, synth.SEED, small=small_, tree_depth=depth, rho=rho_, std=0.05 * emb_scale)
                e_.set_scorer_mode(a.scorer)
                e_.tdm_beam_search_dev(d_seq, U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)
                e_.synchronize(); e_.timing_reset(); barrier()
                nst = max(3, a.steps // 3)
                t0_ = time.perf_counter()
                for i_ in range(nst):
                    e_.tdm_beam_search_dev(d_seqs[i_ % NSH], U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)
                e_.synchronize(); barrier()
                dt_ = max_over_ranks(time.perf_counter() - t0_) / nst
                n0_, k0_ = e_.timing_get_kind(0); n1_, k1_ = e_.timing_get_kind(1)
                ids_ = np.empty((U, a.topk), np.int32); e_.d2h(ids_, d_ids)
                Us_ = min(4096, U)
                tr_ = e_.tdm_beam_search_trace(shard_seqs[(nst - 1) % NSH][:Us_], a.beam, a.topk)
                tc_, tn_ = tr_[3], tr_[5]
                frac_ = []
                for lv_ in range(tn_.shape[1]):
                    if tn_[:, lv_].max() > 0:
                        cand_ = np.concatenate([tc_[u_, lv_, :tn_[u_, lv_]] for u_ in range(Us_)])
                        frac_.append(round(float(np.unique(cand_).size) / float(cand_.size), 4))
                return {"att_scale": att_scale, "emb_scale": emb_scale, "rho": rho_, "users_per_s": world * U / dt_, "ms_per_step": dt_ * 1e3,
                        "kernel": e_.last_beam_kernel(), "kernel_ms": k0_ / max(n0_, 1), "deferred_users_pass_ms": k1_ / max(n1_, 1), "steps": nst,
                        "distinct_result_items": int(np.unique(ids_[ids_ >= 0]).size),
                        "distinct_candidate_rows_per_level_fraction_4096_users": frac_}
            finally:
                e_.close()
        try:
            div_ = _variant(1.7, 32.0, a.rho)
            iid_ = _variant(1.0, 1.0, 0.0)
            diverse = {"workload": "the headline search (same tree, users, beam, topk; device-resident request) on a model whose beams diverge, and on an iid table",
                       "diverse_beams": div_, "iid_table_rho0": iid_,
                       "headline_ms_per_step": dt / a.steps * 1e3,
                       "diverse_slowdown_vs_headline": div_["ms_per_step"] / (dt / a.steps * 1e3),
                       "iid_slowdown_vs_headline": iid_["ms_per_step"] / (dt / a.steps * 1e3),
                       "pmc": "profiles/r05_diverse_summary.json (L2 hit rate 0.667, 129 GB fetched per launch) beside profiles/r05_diverse_head_summary.json (0.968, 11.4 GB)"}
        except Exception as ex:       # noqa: BLE001 — an extra must not cost the headline line
            diverse = {"skipped": repr(ex)}

    if rank == 0:
        avg_ms = kernel_ms / max(n_launch, 1)
        roof = roofline(mode, rows, avg_ms, E, L, kern_name)
        sus_ = sustained_mfma_tflops()
        if sus_.get("f16_random") and "mfma_issued_tflops_f16" in roof:
            # what the matrix pipe SUSTAINS (tools/mfma_power_microbench.hip: a bare MFMA stream, two waves per SIMD, no memory traffic, ~15 ms):
            # the nominal peak is the issue rate at 2.4 GHz; under the switching activity of random operands the chip holds a lower clock
            roof["mfma_sustained_tflops_f16"] = {"quiet_operands": sus_.get("f16_quiet"), "random_operands": sus_["f16_random"],
                                                 "source": "profiles/r05_mfma_power.log (tools/mfma_power_microbench.hip on the same MI355X pool)"}
            roof["issued_frac_of_sustained_random_operand_rate"] = roof["mfma_issued_tflops_f16"] / sus_["f16_random"]
        roof.update({"launches": n_launch, "gather_gbps": rows * (4 * E + 4) / (avg_ms * 1e-3) / 1e9,
                     "second_pass_ms_avg": (defer_ms / n_defer) if n_defer else 0.0,
                     "clock_mhz_under_load": clk_mhz, "socket_power_w_under_load": pw_w, "peak_clock_mhz": 2400.0})
        if clk_mhz:
            # the peaks above are quoted at the 2.4 GHz boost clock; at the clock the kernel is held to, the same work is this fraction
            # of what the pipe can deliver
            roof["frac_at_measured_clock"] = roof["frac"] * 2400.0 / clk_mhz
            if "issued_frac_of_fp16_peak" in roof:
                roof["issued_frac_of_fp16_peak_at_measured_clock"] = roof["issued_frac_of_fp16_peak"] * 2400.0 / clk_mhz
        res = {
            "metric": "beam-search users/sec + recall@200 vs brute-force, 10M-item tree" if a.items == 10_000_000 else
                      "beam-search users/sec (TDM serve, %d-item depth-%d tree, %d-d, beam %d)" % (a.items, depth, E, a.beam),
            "value": world * U * a.steps / dt,
            "unit": "users/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if mode == "f32" else "f32 (operands as fp16 hi+lo pairs on the fp16 matrix pipe, fp32 accumulation)", "data": "synthetic (tree-correlated N(0,0.05) node embeddings rho=%.2f, N(0,0.05) DIN weights, Zipf(1.0) histories)" % a.rho,
            "config": {"workload": "TDM beam-search serving, synthetic %d-item depth-%d binary tree, %d-d emb, "
                                   "beam=%d, topk=%d, L=%d"
                                   % (a.items, depth, E, a.beam, a.topk, L),
                       "items": a.items, "depth": depth, "embed": E, "beam": a.beam, "topk": a.topk, "seq_len": L,
                       "users_per_step_per_gpu": U, "user_shards": NSH, "distinct_users_per_gpu": U * NSH,
                       "parallelism": "user-sharded x%d, replicated table" % world,
                       "scored_rows_per_user": rows / U},
            "roofline": roof,
            "scorer": {"mode": mode, "shift_emb": info["shift_emb"], "shift_w": info["shift_w"]},
        }
        if diverse is not None:
            res["extra_diverse_beams"] = diverse
            if "diverse_beams" in diverse:
                res["diverse_beams_users_per_s"] = diverse["diverse_beams"]["users_per_s"]
        if host_rate is not None:
            res["host_buffer"] = host_rate
            res["host_buffer_users_per_s"] = host_rate["host_buffer_users_per_s"]
        # HBM traffic of the dominant kernel comes from rocprofv3 PMC passes (tools/collect_profiles.sh), which
        # cannot run inside this process; the committed per-launch figure is attached when it was measured on
        # this exact workload, otherwise traffic stays null.
        res["roofline"]["traffic"] = None
        res["roofline"]["algorithmic_gather_bytes_per_launch"] = rows * (4 * E + 4)
        why = None
        for tag in ("r06", "r05", "r04", "r03"):        # newest committed profile set first
            pname = "%s_summary.json" % tag if mode != "f32" else "%s_f32_summary.json" % tag
            try:
                prof = json.load(open(os.path.join(ROOT, "profiles", pname)))
            except Exception:
                continue
            try:
                pw = (prof.get("bench_lines_under_profiler") or [prof.get("bench_line_under_profiler")])[0]["config"]
                pms = prof["kernel_trace"]["avg_ns"] / 1e6
                # the profile is only quoted for the kernel that was just timed, on this workload, and when its own average
                # duration agrees with this run's HIP-event average to 5 % (another kernel version or box state would not)
                if not pw["workload"].startswith(res["config"]["workload"]) or pw["users_per_step_per_gpu"] != U:
                    why = "profiles/%s was taken on another workload" % pname
                elif kern_name not in prof["kernel_trace"]["kernel"]:
                    why = "profiles/%s profiled %s, this run timed %s" % (pname, prof["kernel_trace"]["kernel"], kern_name)
                elif abs(pms - avg_ms) > 0.07 * avg_ms:            # (box-to-box spread of this kernel under the power cap is +-3 % around the profiled box)
                    why = "profiles/%s: profiled kernel average %.2f ms differs from this run's %.2f ms by more than 7 %%" % (pname, pms, avg_ms)
                else:
                    res["roofline"]["traffic"] = prof["hbm_traffic_per_launch_bytes"]["total_corrected"]
                    res["roofline"]["traffic_source"] = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes per launch, "
                                                         "FETCH x2 gfx950 correction (MI355X_MICROARCH.md); attached because the profiled kernel is the "
                                                         "one timed here and its average duration agrees within 7 %%") % pname
                    res["roofline"]["profiled_kernel_ms_avg"] = pms
                    if "mfma_pipe_utilisation" in prof:
                        res["roofline"]["profiled_mfma_busy_frac"] = prof["mfma_pipe_utilisation"]
                    why = None
                    break
            except Exception as ex:       # noqa: BLE001
                why = "profiles/%s unreadable: %r" % (pname, ex)
        if res["roofline"]["traffic"] is None:
            res["roofline"]["traffic_source"] = "null: " + (why or "no committed PMC profile for this kernel")
        if a.recall_users > 0:
            nr = min(a.recall_users, U)
            bids, bsc, bcnt = eng.tdm_bruteforce_topk(seqs[:nr], a.topk)
            rec = [len(set(ids[u, :cnt[u]].tolist()) & set(bids[u, :bcnt[u]].tolist())) / float(a.topk) for u in range(nr)]
            res["recall_at_%d_vs_bruteforce" % a.topk] = {"value": float(np.mean(rec)), "users": nr,
                                                        "definition": "|beam top-k  ∩  brute-force top-k| / k, same scorer weights"}
        table_bytes = (num_index * E + 3 * E * E + 2 * E + 1) * 4
        if world == 1 and a.cpu_users != 0 and host_mem_available() < 2 * table_bytes + (8 << 30):
            res["cpu_baseline_note"] = ("skipped on this catalogue: the oracle needs a %.1f GB host copy of the table and the host "
                                        "reports %.1f GB available; see extra_1m_item_tree.cpu_baseline" %
                                        (table_bytes / 1e9, host_mem_available() / 1e9))
        elif world == 1 and a.cpu_users != 0:
            n_cpu = a.cpu_users
            w = eng.download_weights()
            base, one, outs = cpu_baseline(tree, w, E, L, num_index, seqs, a.beam, a.topk, n_cpu)
            res["cpu_baseline"] = base
            res["cpu_baseline_1core"] = one
            oids, ocnt, otree_, odin_ = outs
            eq = np.array([bool(cnt[u] == ocnt[u] and np.array_equal(ids[u, :cnt[u]], oids[u, :ocnt[u]])) for u in range(len(ocnt))])
            res["cpu_baseline"]["identical_id_lists"] = "%d/%d" % (int(eq.sum()), len(ocnt))
            res["cpu_baseline"]["differing_users"] = int((~eq).sum())
            res["cpu_baseline"]["near_tie"] = near_tie_report(eng, otree_, odin_, seqs[:len(ocnt)][~eq], a.beam, a.topk)
            # top-level scalars (the driver's parsed line keeps scalars): every user whose end-to-end id list differs from the CPU oracle's
            # was traced, and this many of them diverge first at a cut whose score gap is inside the stated tolerance
            nt_ = res["cpu_baseline"]["near_tie"]
            res["id_lists_identical_to_cpu_oracle"] = int(eq.sum())
            res["id_lists_compared_with_cpu_oracle"] = int(len(ocnt))
            res["differing_users_analysed"] = int(nt_["differing_users_analysed"])
            res["differing_users_explained_by_near_tie"] = int(nt_["explained_by_near_tie"])
            res["near_tie_explained_frac"] = (nt_["explained_by_near_tie"] / float(nt_["differing_users_analysed"])) if nt_["differing_users_analysed"] else 1.0
            if other is not None:
                res["cpu_baseline"]["near_tie_between_device_arithmetics"] = near_tie_report(
                    eng, otree_, odin_, seqs[~same_rows], a.beam, a.topk, modes=("split_f16", "f32"))
            del otree_, odin_
        res_main = res
    default_cfg = (a.items, a.depth) == (10_000_000, 24)
    stage("extras on the SAME 10M-item engine OTM serving (BASELINE configs[2], c")
    # ---- extras on the SAME 10M-item engine: OTM serving (BASELINE configs[2], complete-tree variant of the level loop) and the
    #      JTM re-assignment scoring step (configs[3]); host-buffer entry points, so these rates include the PCIe copies ----
    otm = jtm = None
    if a.small and default_cfg:
        Uo = min(32768, U)
        lut = np.zeros(int(tree["leaf_ids"].max()) + 1, np.int32)
        lut[tree["leaf_ids"]] = tree["leaf_codes"]
        ocodes = np.where(seqs[:Uo] > 0, lut[np.clip(seqs[:Uo], 0, lut.size - 1)], -1).astype(np.int32)   # items -> leaf nodes (OTM.scala:15)
        eng.otm_beam_search(ocodes[:1024], a.beam, depth)
        d_os = eng.dev_alloc(Uo * L * 4); d_oi = eng.dev_alloc(Uo * 2 * a.beam * 4); d_osc = eng.dev_alloc(Uo * 2 * a.beam * 4); d_oc = eng.dev_alloc(Uo * 4)
        eng.h2d(d_os, ocodes)
        eng.otm_beam_search_dev(d_os, Uo, L, a.beam, depth, d_oi, d_osc, d_oc)
        sync(); eng.timing_reset(); barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.otm_beam_search_dev(d_os, Uo, L, a.beam, depth, d_oi, d_osc, d_oc)
        sync(); barrier()
        dto = max_over_ranks(time.perf_counter() - t0)
        oout = (np.empty((Uo, 2 * a.beam), np.int32), np.empty((Uo, 2 * a.beam), np.float32), np.empty(Uo, np.int32))
        eng.otm_beam_search(ocodes, a.beam, depth, out=oout)             # (first touch of the result buffers)
        t0 = time.perf_counter()
        oid, osc, ocnt = eng.otm_beam_search(ocodes, a.beam, depth, out=oout)      # host buffers: PCIe copies of 3.3 KB per user included
        dth = time.perf_counter() - t0
        for d_ in (d_os, d_oi, d_osc, d_oc):
            eng.dev_free(d_)
        otm = {"workload": "OTM beam-search serving on the same table: complete depth-%d tree, beam=%d, %d leaf-level candidates per user "
                           "returned; request and results resident in HBM (dm_otm_beam_search_dev)" % (depth, a.beam, 2 * a.beam),
               "users_per_s": world * Uo * 3 / dto, "host_buffer_users_per_s": Uo / dth, "users_per_call": Uo,
               "scorer": eng.scorer_mode()["mode"]}
        # JTM: one gap step (levels 10 -> 12) of TreeLearning.aggregateWeights for a slice of items, 4 training rows each
        from dismember_amd.jtm import JTM
        ni_j = 50_000
        jrng = np.random.default_rng(synth.SEED + 55 + rank)
        jitems = tree["leaf_ids"][:ni_j]
        jrows = {int(it): seqs[jrng.integers(0, a.users, 4)].reshape(-1) for it in jitems}
        jt = JTM(eng, jitems, tree["leaf_codes"][:ni_j], depth, jrows, gap=2, seq_len=L)
        node10 = JTM.ancestor_at_level(jt.item_code, 10)
        jt.child_weights(node10[:], 10, 12)
        sync(); barrier()
        t0 = time.perf_counter()
        wj = jt.child_weights(node10, 10, 12)
        sync(); barrier()
        dtj = max_over_ranks(time.perf_counter() - t0)
        jtm = {"workload": "JTM re-assignment scoring, one gap-2 step (levels 10 -> 12): %d items x 4 training rows x 6 chain nodes "
                           "= %d DIN rows per worker; pairs expanded, scored and summed (reference-order fp32) on the device, host buffers in and out" % (ni_j, ni_j * 4 * 6),
               "items_per_s": world * ni_j / dtj, "din_rows_per_s": world * ni_j * 24 / dtj, "ms": dtj * 1e3,
               "finite_weights": bool(np.isfinite(wj).all())}
    stage("extra the FULL JTM.optimize over the catalogue (BASELINE configs[3]: e")
    # ---- extra: the FULL JTM.optimize over the catalogue (BASELINE configs[3]: every gap step from the root to the leaves, scoring on
    #      the device, greedy re-balance per parent node), items sharded over the ranks when there are several ----
    jtm_full = None
    if a.small and default_cfg and a.jtm_full:
        from dismember_amd.jtm import JTM
        order = np.argsort(tree["leaf_ids"], kind="stable")
        items_s = tree["leaf_ids"][order]; codes_s = tree["leaf_codes"][order]
        nrow = int(a.jtm_rows)
        t0 = time.perf_counter()
        # the catalogue's training rows: the same on every rank (a sharded run splits ONE re-assignment over the ranks)
        hist = synth.make_users(tree["leaf_ids"], min(a.users, 262144), L, np.random.default_rng(synth.SEED + 78))
        pick = np.random.default_rng(synth.SEED + 77).integers(0, len(hist), size=items_s.size * nrow)
        row_ids = hist[pick].reshape(-1)
        row_off = np.arange(items_s.size + 1, dtype=np.int64) * nrow
        jtm_comm, jtm_comm_err = None, None
        if dist is not None:           # N > 1: ONE JTM.optimize sharded over the ranks inside the library (RCCL all-gathers over xGMI)
            if comm is None:
                comm, comm_transport, jtm_comm_err = make_comm(dist, rank, world, local)
                jtm_comm_err_top = jtm_comm_err
            jtm_comm = comm
        jtf = JTM.from_arrays(eng, items_s, codes_s, depth, row_off, row_ids, gap=2, seq_len=L, comm=jtm_comm)
        prep = time.perf_counter() - t0
        sync(); barrier()
        tim = {}
        eng.timing_reset()
        t0 = time.perf_counter()
        projf = jtf.optimize(timing=tim, as_array=True)
        sync(); barrier()
        dtf = max_over_ranks(time.perf_counter() - t0)
        n_rl, rows_ms = eng.timing_get_kind(30)         # the general-rows scorer's launches (HIP events on the library's stream)
        first_leaf = (1 << depth) - 1
        steps_j = (depth + 1) // 2
        din_rows = int(items_s.size) * nrow * 6 * steps_j
        sharded = jtm_comm is not None
        njobs = 1 if sharded else world            # without a communicator every rank runs the whole re-assignment (replicas)
        jtm_full = {"workload": "JTM.optimize, %d items x %d training rows, depth-%d tree, gap 2: %d gap steps, %d DIN rows scored "
                                "(6 chain nodes per row and step); %s"
                                % (items_s.size, nrow, depth, steps_j, din_rows,
                                   ("ONE re-assignment sharded over %d GPUs inside the library (dm_jtm_optimize_cached with a communicator: item-sharded "
                                    "scoring, in-place all-gather of the weight slices, parent-sharded re-balance + all-gather of the projection)" % world)
                                   if sharded else ("single GPU" if world == 1 else "every rank runs the whole re-assignment on its own GPU (replicas: no communicator, %s)" % jtm_comm_err)),
                    "scaling": "strong" if sharded else "weak",
                    "seconds": dtf, "items_per_s": njobs * items_s.size / dtf, "din_rows_per_s": njobs * din_rows / dtf,
                    "scoring_s": tim.get("scoring_s"), "rebalance_s": tim.get("rebalance_s"), "host_glue_s": tim.get("host_glue_s"),
                    "rebalance": "on the device (dm_jtm_optimize_cached: weights and projection stay in HBM)" if tim.get("fused_step_s") else "host",
                    "rows_upload_s": tim.get("rows_upload_s"),
                    "rows_uploaded_by_this_rank": tim.get("rows_uploaded"), "rows_total": int(jtf.row_off[-1]),     # a rank uploads its item range's rows only
                    "synthetic_catalogue_generation_s": prep,
                    "note_on_preparation": "`seconds` covers the whole JTM.optimize call: the upload of the catalogue's training rows (rows_upload_s; the "
                                           "per-row history codes are built on the device once per call), twelve gap steps of scoring + re-balance, the download "
                                           "of the projection.  synthetic_catalogue_generation_s is this bench drawing 40 M synthetic training rows with numpy "
                                           "(round 4 reported it as host_preparation_s): a caller brings its rows, the library never runs it",
                    "bijection_onto_leaves": bool(np.unique(projf).size == projf.size and int(projf.min()) >= first_leaf)}
        if n_rl:
            # the scorer of config 4: dm_din_rows_split_l_kernel<E, L>.  Per scored row 2(2LE + 2E^2 + E) algorithmic flops (merged M = W1b att.W),
            # 192 fp16 MFMAs per 16-row tile issued (3 per product), and (1 + L) embedding rows gathered — 1 + L / chain unique ones, since the
            # 2^(gap+1) - 2 chain nodes of a training row are scored against one history.  Neither pipe bounds it: the fractions say how far.
            rows_l = (int(tim["sharding"]["items_scored"]) if tim.get("sharding") else int(items_s.size) * steps_j) * nrow * 6
            t_ = rows_ms * 1e-3
            fl_ = 2 * (2 * L * E + 2 * E * E + E)
            iss_ = 2 * 3 * (E // 32) * (E // 16) * 16384 / 16.0
            uniq_ = rows_l * (1.0 + L / 6.0) * 4 * E
            jtm_full["roofline"] = {"bound": "l2", "kernel": "dm_din_rows_split_l_kernel<%d, %d>" % (E, L), "launches": n_rl, "kernel_ms_avg": rows_ms / n_rl,
                                    "kernel_ms_total": rows_ms, "rows": rows_l,
                                    "achieved": uniq_ / t_ / 1e9, "peak": 9500.0, "unit": "GB/s", "frac": uniq_ / t_ / 1e9 / 9500.0, "traffic": None,
                                    "peak_source": "tools/gather_microbench.hip: what the L1 / L2 fabric delivers to 1 024 gathering waves (DESIGN.md)",
                                    "unique_row_bytes": uniq_, "requested_gather_gbps": rows_l * (1 + L) * 4 * E / t_ / 1e9,
                                    "algorithmic_tflops": rows_l * fl_ / t_ / 1e12, "mfma_issued_tflops_f16": rows_l * iss_ / t_ / 1e12,
                                    "issued_frac_of_fp16_peak": rows_l * iss_ / t_ / 1e12 / PEAK_MFMA_F16_TFLOPS}
            try:
                pj_ = json.load(open(os.path.join(ROOT, "profiles", "r06_jtm_summary.json")))
                if "dm_din_rows_split_l_kernel" in pj_["kernel_trace"]["kernel"] and abs(pj_["kernel_trace"]["avg_ns"] / 1e6 - rows_ms / n_rl) < 0.15 * rows_ms / n_rl:
                    jtm_full["roofline"]["traffic"] = pj_["hbm_traffic_per_launch_bytes"]["total_corrected"] * n_rl
                    jtm_full["roofline"]["profiled_kernel_ms_avg"] = pj_["kernel_trace"]["avg_ns"] / 1e6
                    jtm_full["roofline"]["profiled_mfma_busy_frac"] = pj_.get("mfma_pipe_utilisation")
            except Exception:       # noqa: BLE001 — the PMC summary is an annotation
                pass
        sh = tim.get("sharding")
        if sh is not None:
            # what THIS run did, from the library's own counters (rank 0's view; per-rank item counts differ by at most one)
            jtm_full["sharding"] = {"rccl_nranks": sh["nranks"] if sh["transport"] == "rccl" else 0, "nranks": sh["nranks"], "transport": sh["transport"],
                                    "items_scored_by_this_rank_per_step": sh["items_scored"] // steps_j,
                                    "items_rebalanced_by_this_rank_in_sharded_steps": sh["items_rebalanced_sharded"],
                                    "steps_replicated_rebalance": sh["steps_replicated_rebalance"], "steps_node_sharded": sh["steps_node_sharded"],
                                    "weight_bytes_gathered": sh["weight_bytes_gathered"], "projection_bytes_gathered": sh["projection_bytes_gathered"],
                                    "exchange_ms": sh["exchange_s"] * 1e3}
        if sharded:
            import zlib
            sums = [None] * world
            dist.all_gather_object(sums, int(zlib.crc32(projf.tobytes())))
            jtm_full["projection_identical_on_all_ranks"] = bool(len(set(sums)) == 1)
            eng.attach_comm(None)
        del jtf, row_ids, pick, projf
    stage("extra BASELINE configs[1] (1M-item depth-20 tree) + one data-parallel ")
    # ---- extra: BASELINE configs[1] (1M-item depth-20 tree) + one data-parallel training step on it ----
    small = train = None
    if a.small and default_cfg:
        for d_ in d_seqs:
            eng.dev_free(d_)
        eng.dev_free(d_ids); eng.dev_free(d_sc); eng.dev_free(d_cnt)
        eng.close()
        depth2, items2 = 20, 1_000_000
        ni2 = (1 << (depth2 + 1)) - 1
        tree2 = synth.make_tree(items2, depth2, np.random.default_rng(synth.SEED))
        seqs2 = synth.make_users(tree2["leaf_ids"], U, L, np.random.default_rng(synth.SEED + 101 + rank))
        eng = Engine(int(os.environ.get("DM_FORCE_DEVICE", local)))
        eng.load_tree(tree2["codes"], tree2["ids"], tree2["is_leaf"], depth2)
        eng.load_id_maps(tree2["leaf_ids"], tree2["leaf_codes"])
        eng.load_weights_din_synthetic(E, ni2, synth.SEED, tree_depth=depth2, rho=a.rho)
        eng.set_scorer_mode(a.scorer)
        d_seq = eng.dev_alloc(U * L * 4); d_ids = eng.dev_alloc(U * a.topk * 4)
        d_sc = eng.dev_alloc(U * a.topk * 4); d_cnt = eng.dev_alloc(U * 4)
        eng.h2d(d_seq, seqs2)
        eng.tdm_beam_search_dev(d_seq, U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)
        sync(); eng.timing_reset(); barrier(); sync()
        t0 = time.perf_counter()
        nst = max(2, a.steps // 2)
        for _ in range(nst):
            eng.tdm_beam_search_dev(d_seq, U, L, a.beam, a.topk, d_ids, d_sc, d_cnt)
        sync(); barrier()
        dt2 = max_over_ranks(time.perf_counter() - t0)
        nl2, kms2 = eng.timing_get_kind(0)
        rows2 = eng.last_scored_rows()
        small = {"workload": "TDM beam-search serving, synthetic %d-item depth-%d tree, %d-d, beam=%d, topk=%d (BASELINE.json configs[1])"
                             % (items2, depth2, E, a.beam, a.topk),
                 "users_per_s": world * U * nst / dt2, "steps": nst, "ms_per_step": dt2 / nst * 1e3,
                 "scored_rows_per_user": rows2 / U,
                 "scorer": eng.scorer_mode()["mode"],
                 "roofline": roofline(eng.scorer_mode()["mode"], rows2, kms2 / max(nl2, 1), E, L, eng.last_beam_kernel())}
        small["roofline"]["traffic_source"] = "null: no PMC pass was committed for this configuration"
        small["roofline_frac"] = small["roofline"]["frac"]
        if rank == 0:
            ids2 = np.empty((U, a.topk), np.int32); cnt2 = np.empty(U, np.int32)
            eng.d2h(ids2, d_ids); eng.d2h(cnt2, d_cnt)
            if a.recall_users > 0:
                nr = min(a.recall_users, U)
                bids, _, bcnt = eng.tdm_bruteforce_topk(seqs2[:nr], a.topk)
                small["recall_at_%d_vs_bruteforce" % a.topk] = float(np.mean(
                    [len(set(ids2[u, :cnt2[u]].tolist()) & set(bids[u, :bcnt[u]].tolist())) / float(a.topk) for u in range(nr)]))
                small["recall_users"] = nr
            if world == 1 and a.cpu_users != 0 and "cpu_baseline" not in res_main:
                base, one, outs = cpu_baseline(tree2, eng.download_weights(), E, L, ni2, seqs2, a.beam, a.topk, a.cpu_users)
                small["cpu_baseline_1core"] = one
                oids, ocnt, otree_, odin_ = outs
                eq = np.array([bool(cnt2[u] == ocnt[u] and np.array_equal(ids2[u, :cnt2[u]], oids[u, :ocnt[u]])) for u in range(len(ocnt))])
                base["identical_id_lists"] = "%d/%d" % (int(eq.sum()), len(ocnt))
                base["differing_users"] = int((~eq).sum())
                base["near_tie"] = near_tie_report(eng, otree_, odin_, seqs2[:len(ocnt)][~eq], a.beam, a.topk)
                del otree_, odin_
                small["cpu_baseline"] = base
        comm_note = None
        if a.train and dist is not None:
            if comm is None:
                comm, comm_transport, comm_err = make_comm(dist, rank, world, local)
            if comm is None:
                comm_note = "training extra skipped: no communicator (%s)" % comm_err
        if a.train and (dist is None or comm is not None):
            from dismember_amd.trainer import TDMTrainer
            neg = np.array(a.params["layer_negative_counts_list"], np.int32)            # model.layer_negative_counts
            per = int(sum(1 + neg[l] for l in range(1, depth2 + 1)))
            Tt = max(1, a.params["total_batch_size"] // per)                # model.total_batch_size expanded rows per worker
            tr = TDMTrainer(eng, neg, lr=a.params["learning_rate"], comm=comm, seed=synth.SEED, sampler="device",
                            with_prob=a.params["sample_with_probability"])
            trng = np.random.default_rng(synth.SEED + 7 + rank)
            tseq = synth.make_users(tree2["leaf_ids"], Tt, L, trng)
            ttgt = trng.choice(tree2["leaf_ids"], Tt).astype(np.int32)
            tr.step(tseq, ttgt)
            sync(); barrier()
            tr.sync_s, tr.sync_calls = 0.0, 0
            t0 = time.perf_counter()
            nts = 5
            for _ in range(nts):
                tloss = tr.step(tseq, ttgt)
            sync(); barrier()
            dtt = max_over_ranks(time.perf_counter() - t0)
            nparam = ni2 * E + 3 * E * E + 2 * E + 1
            train = {"workload": "TDM train step on the 1M-item tree: %d targets -> %d expanded rows per worker (level-wise negatives drawn on the device), "
                                 "DIN fwd+bwd, gradient exchange (dm_train_sync_gradients: RCCL inside the library), Adam over %d parameters "
                                 "(the reference's dense update, evaluated on the rows a gradient has ever reached: bit-identical, dm_adam_last_step_rows)" % (Tt, Tt * per, nparam),
                     "ms_per_step": dtt / nts * 1e3, "rows_per_s": world * Tt * per * nts / dtt, "loss": tloss,
                     "adam_rows_visited_last_step": eng.adam_last_step_rows()[0], "adam_active_rows_path": eng.adam_last_step_rows()[1],
                     "adam_dense_stream_bytes_per_step": 8 * 4 * nparam, "workers": world,
                     "gradient_exchange": ("dm_train_sync_gradients over dm_comm (%s)" % comm_transport) if comm is not None else "single worker"}
            if comm is not None:
                st = eng.train_sync_stats()
                train["exchange"] = {"rccl_nranks": st["nranks"] if st["transport"] == "rccl" else 0, "nranks": st["nranks"], "transport": st["transport"], "ms_per_step": tr.sync_s / max(tr.sync_calls, 1) * 1e3,
                                     "touched_rows_this_rank": st["rows_mine"], "touched_rows_all_ranks": st["rows_total"],
                                     "bytes_sent_per_step": st["bytes_sent"], "bytes_received_per_step": st["bytes_recv"],
                                     "host_syncs_per_step": st["host_syncs"], "rows_per_s_per_rank": Tt * per * nts / dtt}
        elif comm_note:
            train = {"skipped": comm_note}
    stage("extra recall@topk vs brute force on a TRAINED scorer (1M-item tree)")
    # ---- extra: recall@topk vs brute force before / after training at scale (round-5 verdict, next #7).  On random weights the tree is
    # not a max-heap of the scores and the recall says nothing about the search; the reference evaluates trained models
    # (T/evaluation/Evaluator.scala:32-71).  Here: the 1M-item engine, fresh synthetic weights, N TDMTrainer steps (level-wise negatives on
    # the device, DIN fwd+bwd, Adam) on interactions whose targets sit near the user's history in the tree; same held-out users before and after.
    trained = None
    if a.small and default_cfg and a.train and a.trained_recall_steps > 0 and rank == 0:
        try:
            from dismember_amd.trainer import TDMTrainer
            if comm is not None:
                eng.attach_comm(None)
            eng.load_weights_din_synthetic(E, ni2, synth.SEED, tree_depth=depth2, rho=a.rho)
            eng.set_scorer_mode(a.scorer)
            neg = np.array(a.params["layer_negative_counts_list"], np.int32)
            Tn, lr_, spread_ = 1024, 3e-3, 64.0
            tr2 = TDMTrainer(eng, neg, lr=lr_, comm=None, seed=synth.SEED, sampler="device", with_prob=a.params["sample_with_probability"])
            ev_seq, ev_tgt = synth.make_tree_consistent_interactions(tree2["leaf_ids"], 512, L, np.random.default_rng(synth.SEED + 991), spread_)

            def _recall():
                i_, _, c_ = eng.tdm_beam_search(ev_seq, a.beam, a.topk)
                b_, _, bc_ = eng.tdm_bruteforce_topk(ev_seq, a.topk)
                rec_ = np.mean([len(set(i_[u, :c_[u]].tolist()) & set(b_[u, :bc_[u]].tolist())) / float(a.topk) for u in range(len(ev_seq))])
                hit_ = np.mean([int(ev_tgt[u]) in set(i_[u, :c_[u]].tolist()) for u in range(len(ev_seq))])
                bhit_ = np.mean([int(ev_tgt[u]) in set(b_[u, :bc_[u]].tolist()) for u in range(len(ev_seq))])
                return float(rec_), float(hit_), float(bhit_)
            r0_ = _recall()
            trng2 = np.random.default_rng(synth.SEED + 17)
            t0 = time.perf_counter()
            for _ in range(a.trained_recall_steps):
                s_, t_ = synth.make_tree_consistent_interactions(tree2["leaf_ids"], Tn, L, trng2, spread_)
                tl_ = tr2.step(s_, t_)
            eng.synchronize()
            dtr_ = time.perf_counter() - t0
            r1_ = _recall()
            trained = {"what": "recall@%d of the beam search vs brute force under the SAME weights, 512 held-out users, before and after training" % a.topk,
                       "steps": a.trained_recall_steps, "targets_per_step": Tn, "learning_rate": lr_, "leaf_spread": spread_, "train_s": dtr_,
                       "loss_last_step": tl_, "recall_untrained": r0_[0], "recall_trained": r1_[0],
                       "target_in_beam_topk_untrained": r0_[1], "target_in_beam_topk_trained": r1_[1], "target_in_bruteforce_topk_trained": r1_[2],
                       "eval_users": len(ev_seq)}
        except Exception as ex:       # noqa: BLE001 — an extra must not cost the headline line
            trained = {"skipped": repr(ex)}
    barrier()
    stage("extra Deep-Retrieval serving, BASELINE configs[4] (row A13): D=3, K=10")
    # ---- extra: Deep-Retrieval serving, BASELINE configs[4] (row A13): D=3, K=1000, beam=50, 10M items.  The record is the fp64
    #      run (the reference's arithmetic type, deep-retrieval/.../model/LayerModel.scala); the f32 model with the split-fp16
    #      history GEMM is timed beside it on the same inputs ----
    dr = None
    if a.dr and default_cfg:
        eng.close()
        dp = a.dr_params
        Kd, Dd, beam_d, topk_d = int(dp["num_node"]), int(dp["num_layer"]), int(dp["beam_size"]), int(dp["topk_number"])
        Ld, Ed, Jd = int(dp["seq_len"]), int(dp["embed_size"]), int(dp["num_path_per_item"])
        items_d, Ud = 10_000_000, 16384
        drng = np.random.default_rng(synth.SEED + 303 + rank)
        dseq = drng.integers(0, items_d, size=(Ud, Ld)).astype(np.int32)
        dseq[drng.random((Ud, Ld)) < 0.15] = -1
        path_items = synth.dr_path_items_fast(synth.make_dr_paths(items_d, Kd, Dd, Jd, np.random.default_rng(synth.SEED)), Kd)
        nsd = max(2, a.steps // 2)
        runs = {}
        for tag, dtp in (("f64", np.float64), ("f32", np.float32)):
            eng = Engine(int(os.environ.get("DM_FORCE_DEVICE", local)))
            eng.dr_load_model_synthetic(Ed, Ld, Kd, Dd, items_d, synth.SEED, scale=0.05, rerank=True, dtype=dtp)
            eng.dr_load_path_items(*path_items)
            q_seq = eng.dev_alloc(Ud * Ld * 4); eng.h2d(q_seq, dseq)
            q_paths = eng.dev_alloc(Ud * beam_d * Dd * 4); q_probs = eng.dev_alloc(Ud * beam_d * 8); q_cnt = eng.dev_alloc(Ud * 4)
            q_ids = eng.dev_alloc(Ud * topk_d * 4); q_sc = eng.dev_alloc(Ud * topk_d * 8)
            eng.dr_beam_search_dev(q_seq, Ud, beam_d, q_paths, q_probs, q_cnt)
            sync(); eng.timing_reset(); barrier(); sync()
            t0 = time.perf_counter()
            for _ in range(nsd):
                eng.dr_beam_search_dev(q_seq, Ud, beam_d, q_paths, q_probs, q_cnt)
            sync(); barrier()
            dtb = max_over_ranks(time.perf_counter() - t0)
            _, dev_ms_b = eng.timing_get()          # one event pair per search (the library's default): the search's device time
            # breakdown pass, untimed: DM_DR_TIME_LAUNCHES=1 (read per call) puts an event pair around EVERY launch.  The pairs drain the GPU
            # between two kernels (2.12 -> 2.07 ms per 16 384 users without them, 0.63 -> 0.55 per 4 096), which is why the timed loop above runs without.
            os.environ["DM_DR_TIME_LAUNCHES"] = "1"
            try:
                sync(); eng.timing_reset()
                for _ in range(nsd):
                    eng.dr_beam_search_dev(q_seq, Ud, beam_d, q_paths, q_probs, q_cnt)
                sync()
            finally:
                os.environ.pop("DM_DR_TIME_LAUNCHES", None)
            _, kms_b = eng.timing_get()
            # per launch (HIP events on the kernels' stream): kind 0 = the history GEMM, 11 = layer 0, 10 + 2d / 11 + 2d = statistics / cut of layer d
            per_launch = {}
            for kind_, name_ in ((0, "history_gemm"), (11, "layer0"), (12, "stats_layer1"), (13, "cut_layer1"), (14, "stats_layer2"), (15, "cut_layer2"),
                                 (21, "layer0_flagged_users"), (23, "cut_layer1_flagged_users"), (25, "cut_layer2_flagged_users")):
                n_, ms_ = eng.timing_get_kind(kind_)
                if n_:
                    per_launch[name_] = ms_ / nsd
            esz_ = 8 if tag == "f64" else 4
            gemm_ms = per_launch.get("history_gemm", 0.0)
            st_ms = per_launch.get("stats_layer1", 0.0) + per_launch.get("stats_layer2", 0.0)
            # algorithmic work per launch: the history GEMM's 2 D K L E flops per user on the matrix pipe; the statistics kernels' factor rows —
            # per user and layer d: one ES row slice set (K values) + beam x d table rows of K values, read through the XCDs' L2s
            gemm_flops = 2.0 * Dd * Kd * Ld * Ed * Ud
            st_bytes = sum((1 + beam_d * d_) * Kd * esz_ for d_ in range(1, Dd)) * float(Ud)
            peak_mm = 78.6 if tag == "f64" else (2516.6 / 3.0)            # fp64 MFMA; split-fp16: three fp16 MFMAs per product
            rl = {}
            # f32 batches that fill whole rounds of 256 x 256 tiles take the kernel over pre-split operands (dr_kernel.hip.inc: dr_gemm_x_pays, DM_DR_GEMM_X);
            # the batch sizes this extra runs (16 384 and up) always do
            x_tiles = tag == "f32" and Ud >= 12288 and Ed % 64 == 0 and os.environ.get("DM_DR_GEMM_X", "1") != "0" and a.scorer != "f32"
            if gemm_ms > 0:
                rl["roofline"] = {"bound": "mfma", "kernel": "dr_gemm_kernel<double>" if tag == "f64" else ("dr_gemm_split_x_kernel" if x_tiles else "dr_gemm_split_kernel"),
                                  "achieved": gemm_flops / (gemm_ms * 1e-3) / 1e12, "peak": peak_mm, "unit": "TFLOP/s",
                                  "frac": gemm_flops / (gemm_ms * 1e-3) / 1e12 / peak_mm, "kernel_ms_avg": gemm_ms, "traffic": None,
                                  "flops_per_user": 2 * Dd * Kd * Ld * Ed,
                                  "note": "the dominant kernel of the search (%.0f %% of the kernel time)" % (100.0 * gemm_ms / max(kms_b / nsd, 1e-9))}
                sus_ = sustained_mfma_tflops()
                k_ = "f64_random" if tag == "f64" else "f16_random"
                if sus_.get(k_):      # the sustained rate of a bare MFMA stream on random operands (tools/mfma_power_microbench.hip); split-fp16: three MFMAs per product
                    sp_ = sus_[k_] / (1.0 if tag == "f64" else 3.0)
                    rl["roofline"]["peak_sustained_random_operands"] = sp_
                    rl["roofline"]["frac_of_sustained"] = rl["roofline"]["achieved"] / sp_
                # operand requests of the tiled GEMM: every row tile's A operand (gathered history rows) is read once per column tile and every
                # column tile's B operand once per row tile, through the L1 / L2 fabric
                tile_ = 256 if x_tiles else 128
                tiles_m, tiles_n = -(-Ud // tile_), -(-(Dd * Kd) // tile_)
                op_bytes = float(Ud) * Ld * Ed * esz_ * tiles_n + float(Dd * Kd) * Ld * Ed * esz_ * tiles_m
                rl["roofline"]["operand_bytes_per_launch"] = op_bytes
                rl["roofline"]["operand_delivery_TBps"] = op_bytes / (gemm_ms * 1e-3) / 1e12
                rl["roofline"]["operand_delivery_note"] = (("256 x 256 tiles over operands split once at model load, copied global -> LDS directly; knock-out builds "
                                                            "(tools/dr_gemm_probe.hip): matrix work alone 0.23 ms, loads alone 0.19 ms (gather latency, one stage in flight), together 0.33 ms")
                                                           if x_tiles else
                                                           ("128 x 128 tiles: A re-read per column tile, B per row tile; knock-out builds (tools/dr_gemm_probe.hip) put the matrix work "
                                                            "alone at 0.23 ms, the staging alone at 0.31 ms, barely overlapping (VALU split instructions take matrix-pipe issue cycles)")
                                                           if tag == "f32" else
                                                           "128 x 128 tiles, fp64 MFMA: the matrix pipe bounds this kernel (DESIGN.md §3: 0.92 of what its instruction mix delivers)")
                try:          # HBM bytes of the profiled GEMM (rocprofv3 PMC passes of tools/dr_bench.py, profiles/r05_dr_*), attached when the kernel and its duration match
                    prof_ = json.load(open(os.path.join(ROOT, "profiles", "r05_dr_%s_summary.json" % tag)))
                    pk_ = prof_["kernel_trace"]
                    if rl["roofline"]["kernel"].split("<")[0] in pk_["kernel"] and abs(pk_["avg_ns"] / 1e6 - gemm_ms) <= 0.05 * gemm_ms and Ud == 16384:
                        rl["roofline"]["traffic"] = prof_["hbm_traffic_per_launch_bytes"]["total_corrected"]
                        rl["roofline"]["traffic_source"] = "profiles/r05_dr_%s_summary.json (FETCH_SIZE x2 + WRITE_SIZE, separate passes)" % tag
                except Exception:
                    pass
            if st_ms > 0:
                rl["roofline_statistics_kernels"] = {"bound": "l2", "kernel": "drs_stats_kernel<%s, 1> + <%s, 2>" % (("double",) * 2 if tag == "f64" else ("float",) * 2),
                                                     "achieved": st_bytes / (st_ms * 1e-3) / 1e9, "peak": 34500.0, "unit": "GB/s",
                                                     "frac": st_bytes / (st_ms * 1e-3) / 1e9 / 34500.0, "kernel_ms": st_ms, "bytes_per_user": st_bytes / Ud,
                                                     "note": "factor rows read through the eight XCD L2s (aggregate L2 peak, MI355X_MICROARCH.md); the tables' "
                                                             "column slices are L2-resident by construction (88 % TCC hits, profiles/r04_dr_*)"}
            # the three lower bounds of one search step beside the measured time: the history GEMM at the matrix pipe's
            # peak, the table / factor rows at the measured ~9.5 TB/s gather ceiling (tools/gather_microbench.hip), the exps at one per
            # lane and ~25 (f32) / ~60 (f64) VALU instructions each over 1024 SIMDs x 16 lanes/clk
            n_exp = float(sum(2 * Kd for _ in range(1, Dd)) + Kd + 2 * beam_d * Dd) * Ud
            rl["bounds_ms_per_step"] = {"history_gemm_at_mfma_peak": gemm_flops / (peak_mm * 1e12) * 1e3,
                                        "factor_rows_at_9.5_TBps": st_bytes / 9.5e12 * 1e3,
                                        "exps": n_exp * (60 if tag == "f64" else 25) / (1024 * 16 * 2.4e9) * 1e3,
                                        "measured_kernels": kms_b / nsd,
                                        "note": "the sliced search evaluates ~2 K exps per user and layer (history factors) + the winners' exact probabilities "
                                                "instead of one per candidate (beam x K per layer): the exp bound of the one-kernel search was ~%.2f ms"
                                                % (float(sum(beam_d * Kd for _ in range(1, Dd)) + Kd) * Ud * (60 if tag == "f64" else 25) / (1024 * 16 * 2.4e9) * 1e3)}
            paths_h = np.empty((Ud, beam_d, Dd), np.int32)
            eng.d2h(paths_h, q_paths)
            eng.dr_recommend_dev(q_seq, Ud, beam_d, topk_d, q_ids, q_sc, q_cnt)
            sync(); barrier(); sync()
            t0 = time.perf_counter()
            for _ in range(nsd):
                eng.dr_recommend_dev(q_seq, Ud, beam_d, topk_d, q_ids, q_sc, q_cnt)
            sync(); barrier()
            dtr = max_over_ranks(time.perf_counter() - t0)
            runs[tag] = {"beam_search_users_per_s": world * Ud * nsd / dtb, "beam_search_ms_per_step": dtb / nsd * 1e3,
                         "beam_search_kernel_ms_per_step": kms_b / nsd, "beam_search_device_ms_per_step": dev_ms_b / nsd,
                         "timing_note": "beam_search_ms_per_step / users_per_s: wall clock of the timed loop, one event pair per search (device time = beam_search_device_ms_per_step); "
                                        "kernel_ms_by_launch and the rooflines: a second, untimed pass with an event pair around every launch (DM_DR_TIME_LAUNCHES=1), "
                                        "whose kernels do not overlap their neighbours' tails (sum = beam_search_kernel_ms_per_step)", "recommend_users_per_s": world * Ud * nsd / dtr,
                         "kernel_ms_by_launch": per_launch, "paths": paths_h}
            runs[tag].update(rl)
            eng.close()
        eng = Engine(int(os.environ.get("DM_FORCE_DEVICE", local)))       # the later sections expect a live engine to close
        same_paths = int((runs["f64"]["paths"] == runs["f32"]["paths"]).all(axis=(1, 2)).sum())
        for r_ in runs.values():
            del r_["paths"]
        # per user: D history GEMM rows of K x L*E (the only matrix work left) + (1 + beam + beam) table-row sums of K
        dr = {"workload": "Deep-Retrieval serving, D=%d K=%d beam=%d, %d items x %d paths, %d-d (configs/c5_dr_10m.conf), fp64 model and arithmetic "
                          "(the reference's type); f32_split = the same draws as an f32 model, history GEMM in the split-fp16 arithmetic"
                          % (Dd, Kd, beam_d, items_d, Jd, Ed),
              "dtype": "f64", "users_per_step": Ud, "steps": nsd,
              "f32_split": runs["f32"], "f32_users_with_identical_path_lists": "%d/%d" % (same_paths, Ud),
              "gemm_flop_per_user": 2 * Dd * Kd * Ld * Ed, "table_row_bytes_per_user_f64": (beam_d * 1 + beam_d * 2) * Kd * 8,
              "reference_formulation_flop_per_user": 2 * Kd * Ed * (Ld + beam_d * (Ld + 1) + beam_d * (Ld + 2))}
        dr.update(runs["f64"])
        if rank == 0 and a.cpu_users != 0:
            from oracle import pyoracle as po
            small_items = 20000
            wsm = synth.make_dr_model(small_items, Kd, Dd, Ld, Ed, np.random.default_rng(1), scale=0.05)
            orc = po.DeepRetrieval(wsm, Ed, Ld, Kd, Dd, small_items)
            cs = np.random.default_rng(2).integers(0, small_items, size=(24, Ld)).astype(np.int32)
            orc.beam_search(cs[0], beam_d)
            t0 = time.perf_counter()
            for r_ in cs:
                orc.beam_search(r_, beam_d)
            dtc = time.perf_counter() - t0
            dr["cpu_baseline"] = {"value": len(cs) / dtc, "unit": "users/s", "cores": 1, "kind": "port",
                                  "sample": "%d users, fp64 oracle beam search (same D, K, beam, E, L; %d-item catalogue: the work per "
                                            "user does not depend on the catalogue size), 1 thread" % (len(cs), small_items)}
    stage("extra BASELINE configs[2] in the reference's own arithmetic: the OTM s")
    # ---- extra: BASELINE configs[2] in the reference's own arithmetic: the OTM scorer is DIN[Double] (otm/.../model/DIN.scala:12-39).
    #      fp64 model over the complete depth-24 tree (34.4 GB table), serving on the fused fp64 beam kernel, then ONE
    #      LocalOptimizer iteration (otm/.../optim/LocalOptimizer.scala:55-109): pseudo targets, beam nodes, and per level a
    #      forward/backward + gradient exchange + dense fp64 Adam over all 4.29 G parameters ----
    otm64 = None
    if a.otm64 and default_cfg:
        try:
            eng.close()
        except Exception:
            pass
        eng = Engine(int(os.environ.get("DM_FORCE_DEVICE", local)))
        try:
            depth6 = 24
            ni6 = (1 << (depth6 + 1)) - 1
            eng.load_weights_din_synthetic_f64(E, ni6, synth.SEED)
            first6 = (1 << depth6) - 1
            orng = np.random.default_rng(synth.SEED + 606 + rank)
            Uo6 = 16384
            oc6 = (first6 + orng.integers(0, 1 << depth6, size=(Uo6, L))).astype(np.int32)
            oc6[orng.random((Uo6, L)) < 0.15] = -1
            d_s6 = eng.dev_alloc(Uo6 * L * 4); d_i6 = eng.dev_alloc(Uo6 * 2 * a.beam * 4); d_c6 = eng.dev_alloc(Uo6 * 2 * a.beam * 4); d_n6 = eng.dev_alloc(Uo6 * 4)
            eng.h2d(d_s6, oc6)
            eng.otm_beam_search_dev(d_s6, Uo6, L, a.beam, depth6, d_i6, d_c6, d_n6)      # warm-up at full size (first touch of the 34 GB table)
            sync(); eng.timing_reset(); barrier()
            t0 = time.perf_counter()
            calls6 = 3
            for _ in range(calls6):
                eng.otm_beam_search_dev(d_s6, Uo6, L, a.beam, depth6, d_i6, d_c6, d_n6)
            sync(); barrier()
            dt6 = max_over_ranks(time.perf_counter() - t0) / calls6
            nl6, kms6 = eng.timing_get()
            kms6 = kms6 / max(nl6, 1)                      # average per launch
            rows6 = eng.last_scored_rows()                 # of the last call
            fl6 = rows6 * 2.0 * (E * E + 2 * L * E + E)
            otm64 = {"workload": "OTM beam-search serving in fp64 (the reference's DIN[Double]): complete depth-%d tree (%d nodes x %d doubles = %.1f GB), "
                                 "beam=%d, request and results resident in HBM" % (depth6, ni6, E, ni6 * E * 8 / 1e9, a.beam),
                     "scorer": eng.scorer_mode()["mode"], "kernel": eng.last_beam_kernel(), "users_per_call": Uo6, "calls_timed": calls6,
                     "users_per_s": world * Uo6 / dt6, "kernel_ms": kms6, "scored_rows_per_user": rows6 / Uo6,
                     "roofline": {"bound": "mfma", "dtype": "f64", "achieved": fl6 / (kms6 * 1e-3) / 1e12 if kms6 else None, "peak": 78.6,
                                  "unit": "TFLOP/s", "frac": (fl6 / (kms6 * 1e-3) / 78.6e12) if kms6 else None,
                                  "flop_per_row": 2 * (E * E + 2 * L * E + E)}}
            for d_ in (d_s6, d_i6, d_c6, d_n6):
                eng.dev_free(d_)
            # one training iteration in fp64
            from dismember_amd.otm_train import OTMTrainer
            comm6 = None
            if dist is not None:
                if comm is None:
                    comm, comm_transport, comm_err = make_comm(dist, rank, world, local)
                comm6 = comm
            t0 = time.perf_counter()
            tr6, tr6_err = None, None
            try:
                tr6 = OTMTrainer(eng, depth6, a.beam, seq_len=L, lr=1e-4, comm=comm6)      # allocates gradient + Adam state: 103 GB more
                sync()
            except Exception as ex:      # noqa: BLE001
                tr6_err = repr(ex)
            if dist is not None:         # the iteration below is collective: every rank runs it or none does (several ranks squeezed onto
                oks6 = [None] * world    # one GPU for a smoke test do not fit 2 x 155 GB)
                dist.all_gather_object(oks6, tr6_err)
                bad6 = [e_ for e_ in oks6 if e_]
                if bad6:
                    raise RuntimeError("fp64 OTM training extra skipped on every rank: a rank could not set up its trainer (%s)" % bad6[0])
            elif tr6_err:
                raise RuntimeError(tr6_err)
            t_init6 = time.perf_counter() - t0
            Ut6 = 20                                   # 20 users x 400 candidates = 8000 rows per level (model.total_batch_size 8192)
            tseq6 = oc6[:Ut6]
            ttg6 = [(first6 + orng.integers(0, 1 << depth6, size=2)).tolist() for _ in range(Ut6)]
            tr6.train_batch(tseq6, ttg6)
            sync(); barrier()
            t0 = time.perf_counter()
            losses6 = tr6.train_batch(tseq6, ttg6)
            sync(); barrier()
            dtt6 = max_over_ranks(time.perf_counter() - t0)
            npar6 = ni6 * E + 3 * E * E + 2 * E + 1
            otm64["train_iteration"] = {
                "workload": "one OTM LocalOptimizer iteration in fp64: %d users per worker, pseudo targets + beam nodes, then %d levels x "
                            "(forward/backward on %d rows, gradient exchange, fp64 Adam over %d parameters: the dense update on the rows a gradient has ever reached, bit-identical)"
                            % (Ut6, len(losses6), Ut6 * 2 * a.beam, npar6),
                "seconds": dtt6, "levels": len(losses6), "loss_first_level": float(losses6[0]), "loss_last_level": float(losses6[-1]),
                "adam_rows_visited_last_step": eng.adam_last_step_rows()[0], "adam_active_rows_path": eng.adam_last_step_rows()[1],
                "adam_dense_stream_bytes_per_level": 8 * 8 * npar6, "train_init_s": t_init6, "workers": world,
                "gradient_exchange": eng.train_sync_stats() if comm6 is not None else "single worker",
                "phases": tr6.last_stats()}
            # the same iteration at the conf's own batch (configs/c3_otm_10m.conf: model.train_batch_size 8192 USERS per worker, label_num
            # targets each: 3.3 M candidate rows per level), through the same library entry point (dm_otm_train_batch)
            Ub6 = min(8192, Uo6)
            label_num = 5
            bseq6 = oc6[:Ub6]
            btg6 = (first6 + orng.integers(0, 1 << depth6, size=(Ub6, label_num))).tolist()
            tr6.train_batch(bseq6, btg6)
            sync(); eng.timing_reset(); barrier()
            t0 = time.perf_counter()
            lossesb = tr6.train_batch(bseq6, btg6)
            sync(); barrier()
            dtb6 = max_over_ranks(time.perf_counter() - t0)
            ph = tr6.last_stats()
            # the forward/backward's kernels, HIP events around every launch (train_grouped_host.hip.inc): the row kernel is priced on the
            # fp64 MFMAs it ISSUES — per 16-row tile: scores E/4, W1a forward and backward (E/16)^2 x 4 each, dp E/4, the two
            # products over the ceil(L/4) k-steps of the history (E/16) x ceil(L/4) each — x 2048 flops, against the 78.6 TFLOP/s fp64 matrix peak
            kq6 = (L + 3) // 4
            mf_tile = 2 * (E // 4) + 2 * (E // 16) ** 2 * 4 + 2 * (E // 16) * kq6
            nlv, nprev = [], None
            lvl0 = 0
            while (2 << lvl0) <= a.beam:
                lvl0 += 1
            nprev = 1 << lvl0
            for _ in range(len(lossesb)):
                nb_ = nprev if not nlv else min(a.beam, nprev)
                nlv.append(2 * nb_); nprev = 2 * nb_
            tiles6 = sum(Ub6 * ((n_ + 15) // 16) for n_ in nlv)
            kt = {k_: eng.timing_get_kind(k_) for k_ in (40, 41, 42, 43, 44)}
            roof_tr = None
            if kt[41][0] == len(lossesb) and kt[41][1] > 0:
                tf_ = tiles6 * mf_tile * 2048.0 / (kt[41][1] * 1e-3) / 1e12
                mf_wg = tiles6 * ((E // 16) ** 2 + 2 * (E // 16)) * 4            # dW1a + the per-user dG / dK sums, four rows per MFMA
                roof_tr = {"kernel": "tg_rows_kernel<%d, %d>" % (E, kq6), "bound": "mfma", "dtype": "f64", "achieved": tf_, "peak": 78.6,
                           "unit": "TFLOP/s", "frac": tf_ / 78.6, "launches": kt[41][0], "avg_ms": kt[41][1] / kt[41][0],
                           "issued_mfma_per_16_row_tile": mf_tile, "tiles": tiles6,
                           "traffic": None, "traffic_note": "PMC passes of the same iteration: profiles/r05_otmtrain_summary.json",
                           "other_kernels_ms_per_level": {"tg_setup": kt[40][1] / max(kt[40][0], 1), "tg_wgrad": kt[42][1] / max(kt[42][0], 1),
                                                          "tg_user_bwd": kt[43][1] / max(kt[43][0], 1), "dm_wgrad (per-user rows)": kt[44][1] / max(kt[44][0], 1)},
                           "tg_wgrad_frac_of_fp64_peak": (mf_wg * 2048.0 / (kt[42][1] * 1e-3) / 78.6e12) if kt[42][1] else None}
            otm64["train_iteration_batch_8192"] = {
                "workload": "the same iteration at model.train_batch_size = %d users per worker x %d targets: %d levels x %d candidate rows"
                            % (Ub6, label_num, len(lossesb), Ub6 * 2 * a.beam),
                "seconds": dtb6, "users_per_s": world * Ub6 / dtb6, "rows_per_s": world * ph["rows_trained"] / dtb6, "levels": len(lossesb),
                "loss_first_level": float(lossesb[0]), "loss_last_level": float(lossesb[-1]), "roofline": roof_tr,
                "phases": ph, "host_side_s": max(0.0, dtb6 - sum(ph[k] for k in ("pseudo_targets_s", "beam_search_s", "forward_backward_s", "exchange_s", "adam_s"))),
                "adam_rows_visited_last_step": eng.adam_last_step_rows()[0],
                "gradient_exchange": eng.train_sync_stats() if comm6 is not None else "single worker"}
        except Exception as ex:
            otm64 = dict(otm64 or {}, error=repr(ex))
    stage("extra BASELINE configs[0]'s own timer (examples/.../tdm/package.scala:")
    # ---- extra: BASELINE configs[0]'s own timer (examples/.../tdm/package.scala:119-123 prints "Average recommend time"): the bundled
    #      trained E=16 model + tree, one user per call, topk 10, beam 20, 10 warm-up + 100 timed calls through the facade ----
    c1 = None
    if rank == 0 and a.small and default_cfg:
        try:
            from dismember_amd import TDM
            from oracle import pyoracle as po
            gdir = os.path.join(ROOT, "tests", "golden")
            t1 = np.load(os.path.join(gdir, "tdm_tree.npz")); w1 = np.load(os.path.join(gdir, "din_f32.npy"))
            e1 = Engine(int(os.environ.get("DM_FORCE_DEVICE", local)))
            e1.load_tree(t1["codes"], t1["ids"], t1["is_leaf"], int(t1["max_level"])); e1.load_id_maps(t1["leaf_ids"], t1["leaf_codes"])
            e1.load_weights_din(w1, 16, 8191)
            q1 = np.array([0, 0, 2126, 204, 3257, 3439, 996, 1681, 3438, 1882], np.int32)      # the reference's example query
            m1 = TDM(e1, "din")
            for _ in range(10):
                m1.recommend(q1, 10, 20)
            blocks1 = []                     # five blocks of 100 calls, the MEDIAN block reported: late in a long bench run one block can eat a
            for _b in range(5):              # scheduler stall of the whole process (round 6: 0.74 ms per call in one block, 0.06 in every other setting)
                t0 = time.perf_counter()
                for _ in range(100):
                    recs1 = m1.recommend(q1, 10, 20)
                blocks1.append((time.perf_counter() - t0) / 100)
            dt1 = float(np.median(blocks1))
            ot1 = po.TdmTree(t1["codes"], t1["ids"], t1["is_leaf"], t1["leaf_ids"], t1["leaf_codes"], t1["max_level"]); od1 = po.Din(w1, 16, 10, 8191)
            ot1.recommend(od1, q1, 10, 20)
            t0 = time.perf_counter()
            for _ in range(100):
                oi1, _ol = ot1.recommend(od1, q1, 10, 20)
            dto1 = (time.perf_counter() - t0) / 100
            # recall of the beam against brute force on the TRAINED model, the reference's own serving parameters (beam 20, topk 10)
            # and the bench's (beam = topk = 200), 2048 synthetic users over the bundled catalogue
            q1s = synth.make_users(t1["leaf_ids"], 2048, 10, np.random.default_rng(synth.SEED + 909))
            rec1 = {}
            for bm_, tk_ in ((20, 10), (200, 200)):
                bi_, _, bc_ = e1.tdm_beam_search(q1s, bm_, tk_)
                fi_, _, fc_ = e1.tdm_bruteforce_topk(q1s, tk_)
                rec1["beam%d_top%d" % (bm_, tk_)] = float(np.mean([len(set(bi_[u, :bc_[u]].tolist()) & set(fi_[u, :fc_[u]].tolist())) / float(min(tk_, max(1, fc_[u])))
                                                                 for u in range(len(q1s))]))
            c1 = {"workload": "BASELINE configs[0] serving timer: bundled trained E=16 DIN + depth-12 tree (3706 items), TDM.recommend(query, topk=10, "
                              "candidateNum=20), one user per call, 10 warm-up + 100 timed calls",
                  "ms_per_call": dt1 * 1e3, "ms_per_call_blocks_of_100": [b_ * 1e3 for b_ in blocks1], "cpu_oracle_ms_per_call": dto1 * 1e3, "cpu_oracle": "oracle/libdm_oracle.so, 1 thread",
                  "same_items_as_oracle": sorted(r_[0] for r_ in recs1) == sorted(oi1.tolist()),
                  "recall_vs_bruteforce_trained_model": dict(rec1, users=len(q1s),
                                                             definition="|beam top-k  ∩  brute-force top-k| / k on the bundled trained DIN (tests/golden/din_f32.npy)")}
            e1.close()
            # the reference's other two serving timers: OTM (examples/.../otm/package.scala:101-105) on the bundled trained DIN[Double] +
            # item -> leaf mapping, and Deep-Retrieval (examples/.../dr/package.scala:107-111; its bundled model is not in the checkout:
            # a synthetic fp64 model over the bundled item -> path mapping, K = 100, D = 3, E = 16)
            try:
                from dismember_amd import OTM as OTMFacade
                w6 = np.load(os.path.join(gdir, "din_f64.npy")); om = np.load(os.path.join(gdir, "otm_mapping.npy"))
                e2 = Engine(int(os.environ.get("DM_FORCE_DEVICE", local)))
                e2.load_weights_din(w6, 16, 8191)
                mo = OTMFacade(e2, {int(a_): int(b_) for a_, b_ in om})
                qo = [int(x) for x in om[:10, 0]]
                for _ in range(10):
                    mo.recommend(qo, 10, 20)
                t0 = time.perf_counter()
                for _ in range(100):
                    mo.recommend(qo, 10, 20)
                dto2 = (time.perf_counter() - t0) / 100
                od6 = po.Din(w6, 16, 10, 8191)
                qc = np.array([mo.item_id_mapping[i] for i in qo], np.int32)
                po.otm_beam_search(od6, qc, mo.leaf_level, 20)
                t0 = time.perf_counter()
                for _ in range(100):
                    oi6, os6 = po.otm_beam_search(od6, qc, mo.leaf_level, 20)
                    po.otm_finalize(oi6, os6, mo._node_to_item, 10)
                dtoo = (time.perf_counter() - t0) / 100
                c1["otm_recommend"] = {"workload": "OTM.recommend(sequence, topk=10, beamSize=20) on the bundled trained DIN[Double] (fp64 arithmetic on both sides), one user per call, 10 + 100 calls",
                                       "ms_per_call": dto2 * 1e3, "cpu_oracle_ms_per_call": dtoo * 1e3, "kernel": e2.last_beam_kernel()}
                e2.close()
                from dismember_amd import DeepRetrieval as DRFacade
                dm_ = np.load(os.path.join(gdir, "dr_mapping.npz"))
                nit = int(dm_["ids"].max()) + 1
                wdr = synth.make_dr_model(nit, 100, 3, 10, 16, np.random.default_rng(11), scale=0.3)
                ip = np.zeros((nit, dm_["paths"].shape[1], 3), np.int32); ip[dm_["ids"]] = dm_["paths"]
                pit = synth.dr_path_items(ip)
                e3 = Engine(int(os.environ.get("DM_FORCE_DEVICE", local)))
                e3.dr_load_model(wdr, 16, 10, 100, 3, nit, dtype=np.float64)
                e3.dr_load_path_items(*pit)
                md = DRFacade(e3, {int(i_): int(d_) for i_, d_ in zip(dm_["items"], dm_["ids"])})
                qd = [int(x) for x in dm_["items"][:10]]
                for _ in range(10):
                    md.recommend(qd, 10, 20)
                t0 = time.perf_counter()
                for _ in range(100):
                    md.recommend(qd, 10, 20)
                dtd = (time.perf_counter() - t0) / 100
                odr = po.DeepRetrieval(wdr, 16, 10, 100, 3, nit, path_items=pit)
                qdi = np.array([md.item_id_mapping[i] for i in qd], np.int32)
                odr.recommend(qdi, 10, 20)
                t0 = time.perf_counter()
                for _ in range(100):
                    odr.recommend(qdi, 10, 20)
                dtdo = (time.perf_counter() - t0) / 100
                c1["dr_recommend"] = {"workload": "DeepRetrieval.recommend(sequence, topk=10, beamSize=20): synthetic fp64 model (K=100, D=3, E=16) over the bundled "
                                                  "item -> path mapping (%d items x 2 paths), one user per call, 10 + 100 calls" % nit,
                                      "ms_per_call": dtd * 1e3, "cpu_oracle_ms_per_call": dtdo * 1e3}
                e3.close()
            except Exception as ex:
                c1["other_timers_skipped"] = repr(ex)
        except Exception as ex:       # the fixtures are test data: their absence must not break the bench line
            c1 = {"skipped": repr(ex)}
    if rank == 0:
        if dr is not None:
            res_main["extra_deep_retrieval"] = dr
        if c1 is not None:
            res_main["extra_config0_latency"] = c1
        if other is not None:
            res_main["extra_other_scorer"] = other
        if long_hist is not None:
            res_main["extra_long_history"] = long_hist
            res_main["long_history_L24_users_per_s"] = long_hist["fused"]["users_per_s"]
        if otm is not None:
            res_main["extra_otm_serve"] = otm
        if jtm is not None:
            res_main["extra_jtm_scoring"] = jtm
        if jtm_full is not None:
            res_main["extra_jtm_optimize"] = jtm_full
        if otm64 is not None:
            res_main["extra_otm_fp64"] = otm64
        if small is not None:
            res_main["extra_1m_item_tree"] = small
        if train is not None:
            res_main["extra_train_step"] = train
        if trained is not None:
            res_main["extra_trained_recall"] = trained
        if dist is not None:
            res_main["comm_transport"] = comm_transport
            res_main["per_rank_users_per_s"] = per_rank_rate
            nr_ = [x for x in (_dig(train, "exchange", "rccl_nranks"), _dig(jtm_full, "sharding", "rccl_nranks")) if x is not None]
            res_main["rccl_nranks"] = max(nr_) if nr_ else 0
            res_main["comm_error"] = comm_err or jtm_comm_err_top
        print(compact_line(res_main, write_full(res_main)))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if comm is not None:
        comm.close()


if __name__ == "__main__":
    main()
