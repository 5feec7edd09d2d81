#!/usr/bin/env python3
"""bench.py -- GF(2^128) sumcheck (round-eval + fold) throughput on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE complete bivariate-product sumcheck over the resident multilinears: for every
round, the round evaluation (accumulate_kernels -> y_1, y_inf), the host's three XORs / two scalar
multiplications for the round polynomial and the next challenge, and the fold (extrapolate_line)
of every multilinear.  Workload at N = 1: BASELINE.json configs[1] -- m = 2 multilinears of 2^24
BinaryField128b elements (512 MiB), one product claim, 24 rounds.  Inputs are generated once
(SplitMix64), uploaded before the timed region and never modified (the
prover's first fold copies, like the reference's PreFold -> PostFold).  Challenges come from a
fixed SplitMix64 stream instead of a Grøstl transcript (host-side protocol code, out of scope).

N > 1: hypercube sharded on the LAST-bound variables (low index bits under High-to-Low binding,
SURVEY.md section 8e): rank g owns global indices = g mod N as one contiguous local array, every
rank runs the same rounds on 2^24 local elements per multilinear (weak scaling: the global
instance has n = 24 + log2 N variables) and the per-round partial (y_1, y_inf) pairs are combined
with ONE RCCL all_gather of 32 bytes per round + local XOR (RCCL has no XOR reduction).

value = (elements of all multilinears on all ranks) * K / wall seconds  [elems/s].
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-vars", type=int, default=24, help="local variables per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="diagnostic: no per-launch hipEvents in the timed region (no roofline block)")
    ap.add_argument("--cpu-n-vars", type=int, default=0, help="size of the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    force_sharded = os.environ.get("BN_FORCE_SHARDED") == "1"  # exercise the multi-GPU code path on one GPU
    # per-round exchange of the ranks' 32-byte partials: "shm" (host shared memory; all ranks on one
    # node, the measured configuration) or "rccl" (one ncclAllGather per round on the device)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    exchange = os.environ.get("BN_EXCHANGE", "shm" if local_world == int(os.environ.get("WORLD_SIZE", "1")) else "rccl")
    if world > 1 or force_sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # diagnostics for a 1-GPU box: BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo runs several ranks on one device
        # (RCCL refuses that; the shared-memory exchange does not need it)
        if os.environ.get("BN_ALL_ON_GPU0") == "1":
            local_rank = 0
        backend = os.environ.get("BN_PG_BACKEND", "nccl")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)

    import binius_amd
    from binius_amd.distributed import ShardedRoundReducer

    n_vars, m = args.n_vars, 2
    n = 1 << n_vars
    log_world = world.bit_length() - 1
    assert 1 << log_world == world, "number of GPUs must be a power of two"

    # ---- inputs: resident in HBM before the timed region
    from binius_amd import synthetic  # SplitMix64 input streams (numpy)

    hal = binius_amd.Context(local_rank, m * n + m * (n // 2) + 4096)
    hal.set_stream(torch.cuda.current_stream().cuda_stream)
    alloc = hal.dev_alloc()
    d_in = []
    for j in range(m):
        # rank g holds the elements with global index = g mod world; as a stream that is simply an
        # independent uniform array per (multilinear, rank)
        host = synthetic.random_b128(0xB1A50000 + j + 0x100 * rank, n)
        s = alloc.alloc(n)
        hal.copy_h2d(host, s)
        d_in.append(s)
        del host
    stream = synthetic.random_scalars(0xC4A1, n_vars + log_world + 1)
    batch_coeff, challenges = stream[0], stream[1:]
    F = binius_amd.HostField
    reducer = ShardedRoundReducer(hal, dist, world) if dist is not None else None

    # The prover loop runs in the compiled C++ host mirror (binius_amd/host/sumcheck.hpp behind
    # libbinius_amd_host.so): per round one accumulate_kernels, two scalar multiplications and one
    # extrapolate_line per multilinear through the C ABI -- what a Rust host would do, without
    # interpreter time between HAL calls.
    from binius_amd._host import SumcheckPlan

    scratch = alloc.alloc(m * (n // 2) + 64)  # folded copies (+ the residual rounds' folded copies)
    d_partial, d_gathered, rccl, shm = 0, 0, None, None
    if reducer is not None and exchange == "shm":
        # the round loop runs exactly as on one GPU (fused kernels, result mailbox); the ranks' partials
        # meet in a shared-memory segment (binius_amd/host/host_capi.cpp bnh_shm_*).  If the segment
        # cannot be set up on any rank, every rank falls back to the RCCL transport.
        from binius_amd._host import ShmExchange

        try:
            shm = ShmExchange(dist, rank, world)
            ok = 1
        except Exception as ex:  # noqa: BLE001
            print("[bench] rank %d: shared-memory exchange unavailable (%s)" % (rank, ex), file=sys.stderr)
            shm, ok = None, 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if shm is not None:
                shm.close()
            shm, exchange = None, "rccl"
        comm = shm
    if reducer is not None and exchange != "shm":
        # the per-round collective is issued from the compiled host loop: ncclAllGather of the 32-byte
        # partial on the context's stream, communicator bootstrapped over the torch process group
        from binius_amd._host import RcclComm

        rccl = RcclComm(dist, rank, world)
        d_partial = reducer.local.data_ptr()
        d_gathered = reducer.gathered.data_ptr()
        comm = reducer

    # the claimed sum (not timed): inner product on the device, combined across ranks
    claim = hal.inner_product(d_in[0], 7, d_in[1])
    if reducer is not None:
        claim = comm.xor_scalars([claim])[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # shared-memory exchange: the residual log2(G) rounds run inside the same compiled call
    in_call_tail = shm is not None and log_world > 0
    plan = SumcheckPlan(hal, n_vars, d_in, scratch, [(0, 1)], [claim], batch_coeff,
                        challenges[: n_vars + (log_world if in_call_tail else 0)], None, d_partial,
                        rccl.handle if rccl else None, world, d_gathered, shm.handle if shm else None, tail_rounds=in_call_tail)
    tail = None
    if reducer is not None and log_world > 0 and not in_call_tail:
        # residual instance after the local rounds: m multilinears of `world` elements (index = rank)
        d_res = [alloc.alloc(world) for _ in range(m)]
        res_scratch = alloc.alloc(m * max(1, world // 2))

    def one_step():
        plan.run()
        if reducer is None or log_world == 0 or in_call_tail:
            return plan.round_coeffs, plan.final_evals
        # last log2(G) rounds: one all_gather of the m local finals, then a tiny local sumcheck
        per_rank = comm.all_gather_scalars(plan.final_evals())
        running = claim
        for r, (c0, c1, c2) in enumerate(plan.round_coeffs()):
            running = F.mul(F.mul(c2, challenges[r]) ^ c1, challenges[r]) ^ c0
        for j in range(m):
            hal.copy_h2d(binius_amd._ffi_ints_to_arr([per_rank[g][j] for g in range(world)]), d_res[j])
        tp = SumcheckPlan(hal, log_world, d_res, res_scratch, [(0, 1)], [running], batch_coeff, challenges[n_vars : n_vars + log_world])
        tp.run()
        return (lambda: plan.round_coeffs() + tp.round_coeffs()), tp.final_evals

    for _ in range(args.warmup):
        one_step()
    # ---- timed region: exactly K steps, barrier + synchronize on both sides
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        get_coeffs, get_finals = one_step()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    # ---- the same K steps once more with a hipEvent pair around every kernel launch (bn_prof_*), for
    # the roofline block.  The events cost about 5 us per launch on the stream (measured: 1.30 -> 1.41 ms
    # per step), which is instrumentation, not the workload -- so they stay out of `value`; the
    # instrumented pass's own wall time is reported next to it as ms_per_step_instrumented.
    prof = {k: (0.0, 0) for k in hal.PROF_CLASSES}
    elapsed_prof = None
    if not args.no_prof:
        barrier()
        hal.prof_begin()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        barrier()
        elapsed_prof = time.perf_counter() - t2
        prof = hal.prof_end()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of what was timed (N = 1: the sumcheck verifier's final check, on the device values)
    # the sumcheck verifier on what was timed (all ranks hold the same transcript):
    # P_r(0) + P_r(1) == running sum every round, product of the final evaluations == last sum
    ok = True
    running = claim
    for r, (c0, c1, c2) in enumerate(get_coeffs()):
        ok = ok and (c0 ^ (c0 ^ c1 ^ c2)) == running
        running = F.mul(F.mul(c2, challenges[r]) ^ c1, challenges[r]) ^ c0
    fa, fb = get_finals()
    ok = ok and F.mul(fa, fb) == running

    total_elems = m * n * world
    value = total_elems * args.steps / elapsed
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- roofline of the dominant kernel, from hipEvent pairs recorded around every launch in the timed region
    # Algorithmic bytes, per launch at r remaining variables:
    #   round evaluation alone  reads 16*m*2^r                                  (round 0 only)
    #   fold + next evaluation  reads 16*m*2^r, writes 8*m*2^r = 24*m*2^r       (rounds 0..n-2, one fused kernel;
    #                           the evaluation consumes the folded values on chip -- no bytes of its own)
    #   fold alone              24*m*2^r                                        (the last fold, r = 1)
    # The ABI decides per launch which kernel runs; the launch counts per class say what actually ran.
    re_ms, re_cnt = prof["round_eval"]
    fo_ms, fo_cnt = prof["fold"]
    fe_ms, fe_cnt = prof["fold_eval"]
    fs_ms, fs_cnt = prof["fold_eval_small"]
    tl_ms, tl_cnt = prof["tail"]
    fused = fe_cnt + fs_cnt > 0
    if fused:
        # rocprof lists two kernel symbols for the fused launches and so does this block: the f largest
        # rounds are k_foldeval9<2> launches, the next g k_foldeval9_small ones (one workgroup per batch:
        # latency-shaped); what remains (if anything) runs inside one resident k_foldeval_tail launch per
        # step (its time includes the host round trips)
        f = fe_cnt // args.steps
        g = fs_cnt // args.steps
        re_bytes = 16 * m * (1 << n_vars) * args.steps
        fe_bytes = sum(24 * m * (1 << r) for r in range(n_vars - f + 1, n_vars + 1)) * args.steps
        fs_bytes = sum(24 * m * (1 << r) for r in range(n_vars - f - g + 1, n_vars - f + 1)) * args.steps
        tl_bytes = sum(24 * m * (1 << r) for r in range(2, n_vars - f - g + 1)) * args.steps
        fold_bytes = 24 * m * 2 * args.steps
    else:
        re_bytes = sum(16 * m * (1 << r) for r in range(1, n_vars + 1)) * args.steps
        fe_bytes = fs_bytes = tl_bytes = 0
        fold_bytes = sum(24 * (1 << r) for r in range(1, n_vars + 1)) * m * args.steps
    kernels = {
        "k_roundeval9(round_eval)": (re_bytes, re_ms, re_cnt),
        "k_extrapolate_line(fold)": (fold_bytes, fo_ms, fo_cnt),
    }
    if fe_cnt:
        kernels["k_foldeval9(fold+round_eval)"] = (fe_bytes, fe_ms, fe_cnt)
    if fs_cnt:
        kernels["k_foldeval9_small(fold+round_eval, <= 2 batches per CU)"] = (fs_bytes, fs_ms, fs_cnt)
    if tl_cnt:
        kernels["k_foldeval_tail(resident, rounds <= 2^12)"] = (tl_bytes, tl_ms, tl_cnt)
    dom = max(kernels, key=lambda k: kernels[k][1])
    b, ms, cnt = kernels[dom]
    achieved = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    roofline = {
        "bound": "hbm",
        "kernel": dom,
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": None,
        "traffic_source": None,
        "launches": cnt,
        "avg_launch_ms": round(ms / cnt, 5) if cnt else None,
        "algorithmic_bytes_per_launch": b // cnt if cnt else None,
    }
    # HBM traffic of the dominant kernel: PMC counters cannot be collected inside this process, so the
    # value comes from the committed counters-only rocprofv3 passes OF THIS COMMAND on this workload
    # (profiles/r01/bench_n24_pmc.json: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, bytes per launch);
    # null for any other workload or kernel
    try:
        pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "bench_n24_pmc.json")))
        sym = {"k_foldeval9(fold+round_eval)": "k_foldeval9<2>", "k_roundeval9(round_eval)": "k_roundeval9"}.get(dom)
        if sym and pmc["workload"] == {"n_vars": n_vars, "multilinears": m} and sym in pmc["kernels"]:
            roofline["traffic"] = pmc["kernels"][sym]["traffic_bytes_per_launch"]
            roofline["traffic_source"] = "profiles/r01/bench_n24_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; bytes per launch)"
    except (OSError, KeyError, ValueError):
        pass
    # the streaming ceiling this box reaches: a device-to-device copy of one multilinear (read + write), next to
    # the 8 TB/s spec the fractions above are quoted against (report both)
    try:
        cp_src, cp_dst = d_in[0], scratch.slice(0, min(scratch.len, d_in[0].len))
        cp_n = cp_dst.len
        best = None
        for _ in range(4):
            hal.sync()
            hal.timer_begin()
            hal.copy_d2d(cp_src.slice(0, cp_n), cp_dst)
            ms_cp = hal.timer_end_ms()
            best = ms_cp if best is None else min(best, ms_cp)
        copy_gbs = 2 * 16 * cp_n / (best * 1e-3) / 1e9
        roofline["measured_copy_GBps"] = round(copy_gbs, 1)
        roofline["frac_of_measured_copy"] = round(achieved / copy_gbs, 4) if copy_gbs > 0 else None
    except Exception as ex:  # noqa: BLE001 -- the ceiling is a side measurement
        roofline["measured_copy_GBps"] = None
        print("[bench] copy ceiling not measured: %r" % (ex,), file=sys.stderr)
    per_kernel = {
        k: {
            "GBps": round(v[0] / (v[1] * 1e-3) / 1e9, 2) if v[1] > 0 else None,
            "frac": round(v[0] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[1] > 0 else None,
            "total_ms": round(v[1], 3),
            "launches": v[2],
        }
        for k, v in kernels.items()
    }

    out = {
        "metric": "GF(2^128) sumcheck-fold elems/sec + achieved HBM GB/s (% of roofline)",
        "value": value,
        "unit": "elems/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_instrumented": (elapsed_prof * 1e3 / args.steps) if elapsed_prof else None,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "gf2^128 (u128 bitwise)",
        "data": "synthetic",
        "config": {
            "workload": "2^%d-var GF(2^128) bivariate-product sumcheck: round-eval + fold every round, m=2 multilinears per GPU"
            % n_vars,
            "n_vars_local": n_vars,
            "n_vars_global": n_vars + log_world,
            "multilinears": m,
            "sharding": ("low index bits (last-bound variables), one 32-byte exchange per round: "
                         + ("host shared memory" if exchange == "shm" else "RCCL all_gather")) if dist is not None else "none",
        },
        "bit_exact_check": bool(ok),
        "roofline": roofline,
        "kernels": per_kernel,
    }

    # ---- CPU baseline, rank 0 only: the same loop (round-eval + fold every round) on the host cores.
    # Reported value = the OPTIMIZED port (oracle/fastcpu_ref.c: arithmetic in the isomorphic POLYVAL field with
    # PCLMULQDQ, OpenMP over all cores -- BASELINE.md section 2's "best available host ISA"), on the full
    # workload when that takes seconds; the scalar CpuLayer-style port is timed beside it on a bounded sample.
    if rank == 0 and world == 1 and dist is None and not args.no_cpu_baseline:
        import oracle  # the CPU ports being timed (the only use of oracle/ in this file)

        # host cores this process may actually use: affinity mask, capped by the cgroup CPU quota (this pool's
        # containers see 256 logical CPUs and are granted 16; 256 spinning OpenMP threads on a 16-CPU quota
        # run 1000x slower than 16)
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                cores = max(1, min(cores, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
        # scalar port: bounded sample
        cn = args.cpu_n_vars or 16
        while True:
            mls = [oracle.random_b128(0xB1A50000 + j, 1 << cn) for j in range(m)]
            c0 = time.perf_counter()
            oracle.bivariate_sumcheck_prove(mls, cn, [(0, 1)], [0], batch_coeff, challenges[:cn], threads=cores)
            dt = time.perf_counter() - c0
            if args.cpu_n_vars or dt > 2.0 or cn >= 22:
                break
            cn += 2 if dt < 0.5 else 1
        scalar = {"value": m * (1 << cn) / dt, "n_vars": cn, "seconds": round(dt, 2)}
        # optimized port: the workload itself (bounded at 2^26 per multilinear)
        fn = min(n_vars, 26)
        best = None
        for _ in range(3):
            mls = [oracle.random_b128(0xB1A50000 + j, 1 << fn) for j in range(m)]
            c0 = time.perf_counter()
            res = oracle.fast_bivariate_sumcheck_prove(mls, fn, [(0, 1)], [0], batch_coeff, challenges[:fn], threads=cores)
            fdt = time.perf_counter() - c0
            if res is None:
                break
            best = fdt if best is None else min(best, fdt)
        single = None
        if best is not None:
            # one thread, the analogue of RAYON_NUM_THREADS=1 (scripts/run_benchmark.py:211), on 2^22 per multilinear
            sn = min(n_vars, 22)
            mls = [oracle.random_b128(0xB1A50000 + j, 1 << sn) for j in range(m)]
            c0 = time.perf_counter()
            oracle.fast_bivariate_sumcheck_prove(mls, sn, [(0, 1)], [0], batch_coeff, challenges[:sn], threads=1)
            single = {"value": m * (1 << sn) / (time.perf_counter() - c0), "n_vars": sn}
        if best is not None:
            out["cpu_baseline"] = {
                "value": m * (1 << fn) / best,
                "unit": "elems/s",
                "cores": cores,
                "kind": "port",
                "sample": "same sumcheck loop (round-eval + fold every round), m=2, n_vars=%d, optimized C port "
                "(POLYVAL-basis PCLMULQDQ arithmetic, OpenMP, basis conversion of the inputs included), %d threads, best of 3: %.3f s"
                % (fn, cores, best),
                "single_thread": single,
                "scalar_port": {"value": scalar["value"], "sample": "scalar tower-recursion C port, n_vars=%d, %d threads, %.2f s" % (cn, cores, scalar["seconds"])},
            }
        else:
            out["cpu_baseline"] = {
                "value": scalar["value"],
                "unit": "elems/s",
                "cores": cores,
                "kind": "port",
                "sample": "same sumcheck loop, m=2, n_vars=%d, scalar C port (host without PCLMULQDQ), %d threads, %.2f s" % (cn, cores, scalar["seconds"]),
            }

    if rank == 0:
        print(json.dumps(out))
    if rccl is not None:
        rccl.destroy()
    if shm is not None:
        shm.close()
    hal.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
