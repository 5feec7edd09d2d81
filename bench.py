"""bench.py -- GF(2^128) sumcheck (round-eval + fold) throughput on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE complete bivariate-product sumcheck over the resident multilinears: for every
round, the round evaluation (accumulate_kernels -> y_1, y_inf), the host's three XORs / two scalar
multiplications for the round polynomial and the next challenge, and the fold (extrapolate_line)
of every multilinear.

Workload: the north-star instance -- m = 2 multilinears of 2^28 BinaryField128b elements (8 GiB), one
product claim, 28 rounds (BASELINE.json: the configuration the target is quoted on; config 5 when
N = 8).  The GLOBAL instance is fixed, so N > 1 is STRONG scaling: the hypercube is sharded on the
LAST-bound variables (low index bits under High-to-Low binding), rank g owns the
global indices = g mod N as one contiguous local array of 2^(28 - log2 N) elements per multilinear --
a shard of the SAME SplitMix64 instance the single-GPU run proves (synthetic.random_b128_shard).
Every round each rank evaluates its shard and the N partial (y_1, y_inf) pairs are combined by ONE collective inside
the round's kernel: the finalizing workgroup stores its 32-byte partial into every peer's device mailbox (hipIpc-mapped
fine-grained memory, xGMI stores on a node) and XORs what the peers stored ("peer", csrc/finalize.hpp); after the local
rounds the N-element residual instance is rebuilt once and the last log2 N rounds run on it.  The RCCL form (one
ncclAllGather of 32 bytes per round on the context's stream + a one-workgroup XOR) and the host-shared-memory form are
timed beside it and reported in `alt_exchange`, with the per-round cost of each exchange over the shard proven alone.
`--n-vars 24` gives
BASELINE.json configs[1].  Inputs are generated once, uploaded before the timed region and never
modified (the prover's first fold copies, like the reference's PreFold -> PostFold).  Challenges
come from a fixed SplitMix64 stream instead of a Groestl transcript (host-side protocol code, out of
scope).

value = (elements of all multilinears of the global instance) * K / wall seconds  [elems/s].
"""
import argparse
import json
import os
import sys
import time

# the host driver of the GPU boxes only supports dmabuf IPC (RCCL / shared device memory across the ranks)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
PMC_FILE = os.path.join("profiles", "r06", "bench_pmc.json")  # tools/pmc_bench.sh: FETCH_SIZE / WRITE_SIZE passes of this command


def csrc_sha16():
    """Fingerprint of the kernel sources the library was built from (binius_amd/csrc, include/): the PMC file records the
    fingerprint of the build its counters were taken on, and a stale file is refused (roofline.traffic = null) instead of
    describing a binary that no longer exists.  (The GPU box has no .git, so a commit id cannot be checked there.)"""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "binius_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "binius_amd", "csrc", "*.hpp"))
                   + glob.glob(os.path.join(ROOT, "binius_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def transcript_digest(coeffs, finals):
    """16-byte digest of a sumcheck transcript (all round polynomials, then the final evaluations; 16 little-endian bytes
    per field element): equal digests = equal transcripts, whatever the number of ranks that produced them."""
    import hashlib

    h = hashlib.blake2b(digest_size=16)
    for rc in coeffs:
        for c in rc:
            h.update(int(c).to_bytes(16, "little"))
    for f in finals:
        h.update(int(f).to_bytes(16, "little"))
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-vars", type=int, default=28, help="variables of the GLOBAL instance (28: north star; 24: BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-claim-groups", action="store_true", help="skip the k = 4, m = 8 claim-group leg reported beside the headline")
    ap.add_argument("--no-prof", action="store_true", help="diagnostic: no per-launch hipEvents in the timed region (no roofline block)")
    ap.add_argument("--no-alt-exchange", action="store_true", help="do not time the shared-memory exchange beside the RCCL one")
    ap.add_argument("--cpu-n-vars", type=int, default=0, help="size of the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    # `python bench.py --gpus N` run bare (no launcher, WORLD_SIZE unset) must still be an N-rank run: the process replaces itself
    # with the launcher the driver would have used (one rank per GPU, rendezvous on 127.0.0.1), so that a scaling run cannot
    # degrade into an N = 1 line.  And a launcher whose world size disagrees with --gpus is refused rather than mislabelled.
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("[bench] --gpus %d without a launcher: re-executing as %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
        sys.stderr.flush()
        os.execv(sys.executable, cmd)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus and os.environ.get("BN_FORCE_SHARDED") != "1":
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: refusing to print a line for the wrong number of GPUs"
                 % (args.gpus, os.environ.get("WORLD_SIZE", "1")))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    force_sharded = os.environ.get("BN_FORCE_SHARDED") == "1"  # exercise the multi-GPU code path on one GPU
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if world > 1 or force_sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # diagnostics for a 1-GPU box: BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo runs several ranks on one device
        # (RCCL refuses that; the shared-memory exchange does not need it)
        if os.environ.get("BN_ALL_ON_GPU0") == "1":
            local_rank = 0
            # ranks that share a device must not arm rounds (csrc/arm.hpp): an armed kernel waits ON the device, and several
            # processes' worth of waiting workgroups leave no compute units for the kernels the challenges depend on
            os.environ.setdefault("BN_ARM", "0")
        backend = os.environ.get("BN_PG_BACKEND", "nccl")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    # per-round exchange of the ranks' 32-byte partials:
    #   "peer"  every rank's finalize step stores its partial into every peer's device mailbox (hipIpc-mapped fine-grained
    #           memory: xGMI stores on a node) and XORs what the peers stored -- ONE hand-written collective per round, inside
    #           the round's kernel; the small rounds stay armed.  The default.
    #   "rccl"  one ncclAllGather per round on the context's stream + a one-workgroup XOR + readback (timed beside it)
    #   "shm"   the partials meet in host shared memory (timed beside it)
    rccl_possible = dist is not None and dist.get_backend() == "nccl"
    # BN_EXCHANGE=auto (the default): every exchange that works on this node is tried for two untimed steps before the
    # warm-up and the fastest one (MAX over ranks) becomes the default; the choice and all trial times are reported.
    exchange = os.environ.get("BN_EXCHANGE", "auto")
    import binius_amd
    from binius_amd import synthetic  # SplitMix64 input streams (numpy)

    # the thread that drives the context runs on the NUMA node the device hangs off (every small round is a host -> device ->
    # host round trip; from the other socket each one crosses the socket interconnect too: 15.0 -> 17.1 us per two-round
    # launch).  Before the context is created, so that its pinned mailboxes are allocated there as well.  BN_BIND_NUMA=0: off.
    host_affinity = "unchanged (BN_BIND_NUMA=0)" if os.environ.get("BN_BIND_NUMA") == "0" else binius_amd.bind_host_thread_to_device(local_rank)
    from binius_amd._host import PeerExchange, RcclComm, ShmExchange, SumcheckPlan

    m = 2
    log_world = world.bit_length() - 1
    assert 1 << log_world == world, "number of GPUs must be a power of two"
    n_global = args.n_vars
    n_vars = n_global - log_world  # local variables per rank
    assert n_vars >= 2, "instance too small for this many ranks"
    n = 1 << n_vars

    # ---- inputs: resident in HBM before the timed region.  The context keeps its OWN stream (deferred
    # folds are legal there, include/binius_amd.h "Stream contract"); RCCL collectives are enqueued on it by
    # the compiled host loop after bn_ctx_get_stream.
    hal = binius_amd.Context(local_rank, m * n + m * (n // 2) + 8 * world + 4096)
    alloc = hal.dev_alloc()
    d_in = []
    chunk = 1 << 22
    for j in range(m):
        s = alloc.alloc(n)
        for off in range(0, n, chunk):
            c = min(chunk, n - off)
            hal.copy_h2d(synthetic.random_b128_shard(0xB1A50000 + j, c, world, rank, start=off), s.slice(off, off + c))
        d_in.append(s)
    stream = synthetic.random_scalars(0xC4A1, n_global + 1)
    batch_coeff, challenges = stream[0], stream[1:]
    F = binius_amd.HostField
    scratch = alloc.alloc(m * (n // 2) + 8 * world + 64)  # folded copies (+ the residual instance and its folded copies)

    shm, rccl, reducer, peer = None, None, None, None
    d_partial, d_gathered = 0, 0
    available = []
    # what happened to every exchange on this node, whatever the default ends up being: the first run on a
    # real multi-GPU node must be diagnostic even if it falls back
    ex_rec = {k: {"set_up": False, "tried": False, "ok": None, "error": None, "trial_ms_per_step": None, "ms_per_step": None,
                  "same_transcript": None, "exchange_us_per_round": None} for k in ("peer", "shm", "rccl")}
    if dist is not None:
        from binius_amd.distributed import ShardedRoundReducer

        flag_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"

        def everybody(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        reducer = ShardedRoundReducer(hal, dist, world)  # device buffers of the RCCL exchange + scalar all-gathers
        # the shared-memory segment: the "shm" exchange, and the one-off rebuild of the residual instance under "peer"
        if local_world == world:
            try:
                shm = ShmExchange(dist, rank, world)
            except Exception as ex:  # noqa: BLE001
                print("[bench] rank %d: shared-memory exchange unavailable (%s)" % (rank, ex), file=sys.stderr)
                ex_rec["shm"]["error"] = "set-up failed on rank %d: %s" % (rank, ex)
                shm = None
            if not everybody(shm is not None):
                if shm is not None:
                    shm.close()
                shm = None
        if shm is not None:
            available.append("shm")
            try:
                peer = PeerExchange(hal, dist, rank, world)  # (collective: every rank learns whether all of them connected)
                available.insert(0, "peer")
            except Exception as ex:  # noqa: BLE001
                print("[bench] rank %d: peer exchange unavailable (%s)" % (rank, ex), file=sys.stderr)
                ex_rec["peer"]["error"] = "set-up failed on rank %d: %s" % (rank, ex)
                peer = None
        if rccl_possible:
            # the per-round collective is issued from the compiled host loop: ncclAllGather of the 32-byte
            # partial on the context's stream, communicator bootstrapped over the torch process group
            try:
                rccl = RcclComm(dist, rank, world)
            except Exception as ex:  # noqa: BLE001
                print("[bench] rank %d: RCCL communicator unavailable (%s)" % (rank, ex), file=sys.stderr)
                ex_rec["rccl"]["error"] = "set-up failed on rank %d: %s" % (rank, ex)
                rccl = None
            if not everybody(rccl is not None):
                if rccl is not None:
                    rccl.destroy()
                rccl = None
            else:
                d_partial = reducer.local.data_ptr()
                d_gathered = reducer.gathered.data_ptr()
                available.append("rccl")
        for k in available:
            ex_rec[k]["set_up"] = True
        if not rccl_possible:
            ex_rec["rccl"]["error"] = "process group backend is %s, not nccl" % dist.get_backend()
        if local_world != world:
            ex_rec["shm"]["error"] = ex_rec["peer"]["error"] = "ranks span several nodes"
        if not available:
            raise SystemExit("no exchange is available on this node (shared memory, peer mailboxes and RCCL all failed): " + json.dumps(ex_rec))
        if exchange != "auto" and exchange not in available:
            exchange = "%s (fallback: %s is not available on this node)" % (available[0], exchange)

    # the claimed sum (not timed): inner product on the device, combined across ranks
    claim = hal.inner_product(d_in[0], 7, d_in[1])
    if dist is not None:
        claim = (shm if shm is not None else reducer).xor_scalars([claim])[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        hal.sync()
        torch.cuda.synchronize()

    def make_plan(kind):
        kind = kind.split(" ")[0]
        if kind == "solo":  # diagnostic: the local shard alone, no exchange, no residual rounds (its transcript means nothing)
            return SumcheckPlan(hal, n_vars, d_in, scratch, [(0, 1)], [claim], batch_coeff, challenges[:n_vars])
        if kind == "rccl":
            return SumcheckPlan(hal, n_vars, d_in, scratch, [(0, 1)], [claim], batch_coeff, challenges[:n_global], None, d_partial, rccl.handle, world,
                                d_gathered, None, tail_rounds=log_world > 0)
        return SumcheckPlan(hal, n_vars, d_in, scratch, [(0, 1)], [claim], batch_coeff, challenges[:n_global], None, 0, None, world, 0, shm.handle,
                            tail_rounds=log_world > 0, peer=(kind == "peer"))

    if dist is None:
        plan = SumcheckPlan(hal, n_vars, d_in, scratch, [(0, 1)], [claim], batch_coeff, challenges[:n_global])
    else:
        plan = make_plan(exchange) if exchange != "auto" else None  # (auto: chosen below)

    def timed(pl, steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            pl.run()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    trials = None
    if dist is not None and exchange == "auto":
        # ---- pick the exchange by measurement (untimed: before the warm-up).  A candidate that raises on any rank is dropped.
        trials = {}
        for cand in list(available):
            pl = make_plan(cand)
            ok = True
            ex_rec[cand]["tried"] = True
            try:
                pl.run()
            except Exception as ex:  # noqa: BLE001
                print("[bench] rank %d: exchange %s failed (%s)" % (rank, cand, ex), file=sys.stderr)
                ex_rec[cand]["error"] = "first run raised on rank %d: %s" % (rank, ex)
                ok = False
            if not everybody(ok):
                available.remove(cand)
                trials[cand] = None
                ex_rec[cand]["ok"] = False
                if ex_rec[cand]["error"] is None:
                    ex_rec[cand]["error"] = "first run raised on another rank"
                continue
            trials[cand] = timed(pl, 2) * 1e3 / 2
            ex_rec[cand]["ok"] = True
            ex_rec[cand]["trial_ms_per_step"] = round(trials[cand], 4)
        if not available:
            raise SystemExit("every exchange failed on this node")
        exchange = min(available, key=lambda c: trials[c])
        plan = make_plan(exchange)
    # First run of the default exchange, guarded: if it cannot run on this node (an error on ANY rank), every rank
    # switches to the next available one together and the line says so -- a scaling line with a documented fallback
    # beats none.
    if dist is not None:
        while True:
            ok = True
            ex_rec[exchange.split(" ")[0]]["tried"] = True
            try:
                plan.run()
            except Exception as ex:  # noqa: BLE001
                print("[bench] rank %d: exchange %s failed (%s)" % (rank, exchange.split(" ")[0], ex), file=sys.stderr)
                ex_rec[exchange.split(" ")[0]]["error"] = "run raised on rank %d: %s" % (rank, ex)
                ok = False
            if everybody(ok):
                ex_rec[exchange.split(" ")[0]]["ok"] = True
                break
            failed = exchange.split(" ")[0]
            ex_rec[failed]["ok"] = False
            if ex_rec[failed]["error"] is None:
                ex_rec[failed]["error"] = "run raised on another rank"
            available.remove(failed)
            if not available:
                raise SystemExit("every exchange failed on this node")
            exchange = "%s (fallback: the %s exchange raised an error on this node)" % (available[0], failed)
            plan = make_plan(exchange)
    for _ in range(args.warmup):
        plan.run()
    # ---- timed region: exactly K steps, barrier + synchronize on both sides, MAX over ranks
    elapsed = timed(plan, args.steps)
    get_coeffs, get_finals = plan.round_coeffs, plan.final_evals
    # ---- the same K steps once more with a hipEvent pair around every kernel launch (bn_prof_*), for
    # the roofline block.  The events cost about 5 us per launch on the stream, which is instrumentation,
    # not the workload -- so they stay out of `value`; the instrumented pass's own wall time is reported
    # next to it as ms_per_step_instrumented.
    prof = {k: (0.0, 0) for k in hal.PROF_CLASSES}
    elapsed_prof = None
    if not args.no_prof:
        barrier()
        hal.prof_begin()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            plan.run()
        barrier()
        elapsed_prof = time.perf_counter() - t2
        prof = hal.prof_end()
    # ---- the other exchanges, same steps, and the local shard with no exchange at all (diagnostic): reported beside the
    # default, never as `value`.  exchange_us_per_round = (step with the exchange - step of the shard alone) / rounds.
    alt = None
    if dist is not None and not args.no_alt_exchange:
        alt = []
        solo = make_plan("solo")
        solo.run()
        dt_solo = timed(solo, args.steps)
        n_rounds = n_global
        main_kind = exchange.split(" ")[0]
        for other in available:
            if other == main_kind:
                continue
            plan_alt = make_plan(other)
            ok_alt = True
            ex_rec[other]["tried"] = True
            try:
                plan_alt.run()
            except Exception as ex:  # noqa: BLE001
                print("[bench] rank %d: exchange %s failed (%s)" % (rank, other, ex), file=sys.stderr)
                ex_rec[other]["error"] = "run raised on rank %d: %s" % (rank, ex)
                ok_alt = False
            if not everybody(ok_alt):
                alt.append({"exchange": other, "error": "failed on this node"})
                ex_rec[other]["ok"] = False
                continue
            dt_alt = timed(plan_alt, args.steps)
            same = plan_alt.round_coeffs() == get_coeffs() and plan_alt.final_evals() == get_finals()
            ex_rec[other].update(ok=True, ms_per_step=round(dt_alt * 1e3 / args.steps, 4), same_transcript=bool(same),
                                 exchange_us_per_round=round((dt_alt - dt_solo) * 1e6 / args.steps / n_rounds, 2))
            alt.append({"exchange": other, "ms_per_step": dt_alt * 1e3 / args.steps, "value": m * (1 << n_global) * args.steps / dt_alt,
                        "same_transcript_as_default": bool(same),
                        "exchange_us_per_round": round((dt_alt - dt_solo) * 1e6 / args.steps / n_rounds, 2)})
        ex_rec[main_kind].update(ms_per_step=round(elapsed * 1e3 / args.steps, 4), same_transcript=True,
                                 exchange_us_per_round=round((elapsed - dt_solo) * 1e6 / args.steps / n_rounds, 2))
        alt.append({"exchange": "none (the local shard alone, no residual rounds: diagnostic)", "ms_per_step": dt_solo * 1e3 / args.steps})
        alt.append({"exchange": main_kind + " (the default: `value`)", "exchange_us_per_round": round((elapsed - dt_solo) * 1e6 / args.steps / n_rounds, 2),
                    "chosen_by": "measurement before the warm-up: ms per step " + json.dumps(trials) if trials is not None else "BN_EXCHANGE"})

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

    total_elems = m * (1 << n_global)
    value = total_elems * args.steps / elapsed
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- roofline of the dominant kernel, from hipEvent pairs recorded around every launch in the timed region
    # Algorithmic bytes, per launch at r remaining variables:
    #   round evaluation alone  reads 16*m*2^r                                  (round 0 only)
    #   fold + next evaluation  reads 16*m*2^r, writes 8*m*2^r = 24*m*2^r       (rounds 0..n-2, one fused kernel;
    #                           the evaluation consumes the folded values on chip -- no bytes of its own)
    #   fold alone              24*m*2^r                                        (the last fold, r = 1)
    # The ABI decides per launch which kernel runs; the launch counts per class say what actually ran.
    # Classes (include/binius_amd.h BN_PROF_*), as rocprof lists the kernel symbols:
    #   round_eval_mfma  k_roundeval_fp4_ws / k_roundeval_fp4 (>= 2^20 points: FP4 matrix path) / k_roundeval_mfma (int8; BN_FP4=0)   round 0 on the matrix cores
    #                                                                          fold_eval_mfma   k_foldeval_mfma_fp4 (whole tiles, >= 2 per CU) / k_foldeval_mfma (the rest down to 2 tiles per CU)
    #   round_eval       k_roundeval9       round 0, 9-lane VALU kernel        fold_eval        k_foldeval9<2>   (the next size down)
    #   fold             k_extrapolate_line / k_fold_publish (last fold)       fold_eval_small  k_foldeval9_small (one workgroup per batch: latency-shaped)
    #                                                                          tail             k_foldeval_tail  (resident; opt-in)
    #                                                                          fold_eval8       k_foldeval8      (two rounds per launch: the small rounds)
    K = args.steps
    counts = {c: prof[c][1] // K if K else 0 for c in prof}
    # the fused launches of a step, largest round first: mfma, then k_foldeval9<2>, then small, then tail
    order = ["fold_eval_mfma", "fold_eval", "fold_eval_small"]
    r_hi = n_vars  # r of the next fused launch (pre-fold size 2^r)
    fused_bytes = {}
    for c in order:
        f = counts.get(c, 0)
        fused_bytes[c] = sum(24 * m * (1 << r) for r in range(r_hi - f + 1, r_hi + 1)) * K
        r_hi -= f
    # two-round launches (k_foldeval8, kernels_foldeval8.hip): the first folds once (pre-fold size 2^r_hi), every later one
    # folds twice -- 2^(r_hi - 1), 2^(r_hi - 3), ... -- each reads its arrays once and writes half of them: 24*m*n_in
    f8 = counts.get("fold_eval8", 0)
    fused_bytes["fold_eval8"] = sum(24 * m * (1 << r) for r in ([r_hi] + [r_hi - 1 - 2 * i for i in range(f8 - 1)] if f8 else []) if r >= 2) * K
    fused = sum(counts.get(c, 0) for c in order) + counts.get("tail", 0) + f8 > 0
    tl_bytes = sum(24 * m * (1 << r) for r in range(2, r_hi + 1)) * K if counts.get("tail", 0) else 0
    n_re = counts.get("round_eval", 0) + counts.get("round_eval_mfma", 0)
    if fused:
        re_bytes_each = 16 * m * (1 << n_vars)  # round 0 (and round 0 of the residual instance: negligible)
        fold_bytes = 24 * m * 2 * K
    else:
        re_bytes_each = None
        fold_bytes = sum(24 * (1 << r) for r in range(1, n_vars + 1)) * m * K
    # round 0 has 2^(n_vars - 1) points: from 2^20 on it runs on the FP4 matrix path (csrc/kernels_roundeval_mfma.hip fp4_applies)
    fp4_round0 = os.environ.get("BN_FP4", "1")[:1] != "0" and n_vars - 1 >= int(os.environ.get("BN_FP4_MIN_LOG2", "20"))
    label = {
        "round_eval_mfma": "k_roundeval_fp4(round_eval)" if fp4_round0 else "k_roundeval_mfma(round_eval)",
        "round_eval": "k_roundeval9(round_eval)",
        "fold": "k_extrapolate_line(fold)",
        "fold_eval_mfma": "k_foldeval_mfma(fold+round_eval)",
        "fold_eval": "k_foldeval9(fold+round_eval)",
        "fold_eval_small": "k_foldeval9_small(fold+round_eval, <= 2 batches per CU)",
        "fold_eval8": "k_foldeval8(two rounds per launch: 1-2 folds + the eight quarter sums; the small rounds)",
        "tail": "k_foldeval_tail(resident, rounds <= 2^12)",
    }
    kernels = {}
    for c in ("round_eval_mfma", "round_eval"):
        ms_c, cnt_c = prof[c]
        if cnt_c:
            if fused:
                # the large launch of the class is round 0 of the local instance; residual-instance launches are bytes-free here
                bytes_c = re_bytes_each * K if (c == "round_eval_mfma" or counts.get("round_eval_mfma", 0) == 0) else 0
            else:
                bytes_c = sum(16 * m * (1 << r) for r in range(1, n_vars + 1)) * K
            kernels[label[c]] = (bytes_c, ms_c, cnt_c)
    kernels[label["fold"]] = (fold_bytes, prof["fold"][0], prof["fold"][1])
    for c in order + ["fold_eval8"]:
        if prof[c][1]:
            kernels[label[c]] = (fused_bytes[c], prof[c][0], prof[c][1])
    if prof["tail"][1]:
        kernels[label["tail"]] = (tl_bytes, prof["tail"][0], prof["tail"][1])
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
    # value comes from the committed counters-only rocprofv3 passes OF THIS COMMAND (tools/pmc_bench.sh ->
    # profiles/r06/bench_pmc.json: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, average bytes per launch
    # of the kernel symbol, with the commit of the build it was taken on); null for a workload it does not hold
    try:
        pmc = json.load(open(os.path.join(ROOT, PMC_FILE)))
        sym = dom.split("(")[0]
        ent = pmc["workloads"].get("n_vars_local=%d,m=%d" % (n_vars, m), {}).get(sym)
        if ent and pmc.get("csrc_sha16") != csrc_sha16():
            roofline["traffic_source"] = "%s is stale: taken on kernel sources %s, this tree is %s" % (PMC_FILE, pmc.get("csrc_sha16"), csrc_sha16())
        elif ent:
            roofline["traffic"] = ent["traffic_bytes_per_launch"]
            roofline["traffic_source"] = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at build %s; average bytes per launch of %s)" % (
                PMC_FILE, pmc.get("build", "?"), sym)
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
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "gf2^128 (u128 bitwise)",
        "data": "synthetic",
        "config": {
            "workload": "2^%d-var GF(2^128) bivariate-product sumcheck: round-eval + fold every round, m=2 multilinears, %s"
            % (n_global, "1 GPU" if world == 1 else "sharded on the top-bound-last (low index) variables across %d GPUs" % world),
            "n_vars_local": n_vars,
            "n_vars_global": n_global,
            "multilinears": m,
            "host_affinity": host_affinity,
            "sharding": ("low index bits (last-bound variables), one 32-byte exchange per round: "
                         + {"peer": "device mailboxes (every rank's finalize step stores its partial into every peer's hipIpc-mapped mailbox and XORs the world's)",
                            "shm": "host shared memory", "rccl": "RCCL all_gather on the context's stream + device XOR"}[exchange.split(" ")[0]]
                         + (" -- " + exchange if " (" in exchange else "")) if dist is not None else "none",
            # north_star names RCCL for the cross-device reduce.  It is set up, run and timed on every multi-GPU
            # invocation (`exchanges.rccl`, `alt_exchange`), and it is the default only if it is the fastest of the three there:
            # its result lives on the device (all_gather on the stream + a one-workgroup XOR), which rules out everything that
            # takes a small round off the launch path -- armed rounds, two-round launches, the host tail -- and
            # costs a collective launch per round where the other two exchange 32 bytes inside the round's own kernel.
            "named_collective": (None if dist is None else
                                 "rccl is the default exchange of this run" if exchange.split(" ")[0] == "rccl" else
                                 "rccl (north_star's collective) timed beside the default in `exchanges`; not the default: a device-resident "
                                 "result excludes armed / two-round / host-tail rounds and adds a collective launch per round"),
        },
        "alt_exchange": alt,
        # every exchange on this node: was it set up, tried, did it run on all ranks (error text if not), its ms per step and
        # whether it produced the default's transcript -- filled in even when the default is a fallback
        "exchanges": ex_rec if dist is not None else None,
        # the sumcheck verifier's equations on the device-produced transcript (claim from the device inner product):
        # P_r(0) + P_r(1) == running sum every round, product of the final evaluations == last sum.  Bit-exact parity of
        # this very instance against the CPU oracle is tests/test_gpu_north_star.py (n = 24, n = 28, 8 shards).
        "verifier_check": bool(ok),
        "transcript_digest": transcript_digest(get_coeffs(), get_finals()),
        "roofline": roofline,
        "kernels": per_kernel,
    }

    # ---- the reference's REAL call shape beside the headline: ONE BivariateSumcheckProver with k product claims over m multilinears
    # (piop::prove builds one such prover per size out of EVERY committed multilinear of that size and its transparents,
    # core/src/piop/prove.rs:262-287), through the same compiled prover loop -- the claim-group path (csrc/abi_group.cpp,
    # kernels_group.hip).  Two shapes: k = 4 / m = 8 at 2^24 (round 5's), and the keccak table's width, k = 50 / m = 100 at 2^22
    # (m3/src/gadgets/hash/keccak/stacked.rs:105,292).  Reported, not the metric.
    def claim_groups_leg(gk, gn_vars):
        gm, gn = 2 * gk, 1 << gn_vars
        with binius_amd.Context(local_rank, gm * gn + gm * (gn // 2) + 4096) as ghal:
            galloc = ghal.dev_alloc()
            gd = []
            for j in range(gm):  # dense pseudo-random inputs generated on the device: tensor expansions of random points
                s = galloc.alloc(gn)
                ghal.fill(s.slice(0, 1), 1 + j)
                ghal.tensor_expand(0, synthetic.random_scalars(0xB1A5 + j, gn_vars), s)
                gd.append(s)
            gcomps = [(i, gk + i) for i in range(gk)]
            gsums = [ghal.inner_product(gd[i], 7, gd[j]) for i, j in gcomps]
            gstream = synthetic.random_scalars(0xC4A1, gn_vars + 1)
            gplan = SumcheckPlan(ghal, gn_vars, gd, galloc.alloc(gm * (gn // 2)), gcomps, gsums, gstream[0], gstream[1:])
            gplan.run()
            ghal.sync()
            c_before = ghal.group_counters()
            t0 = time.perf_counter()
            for _ in range(3):
                gplan.run()
            ghal.sync()
            gms = (time.perf_counter() - t0) * 1e3 / 3
            cnt = ghal.group_counters()
            gc, gf = gplan.round_coeffs(), gplan.final_evals()
            run_sum, p = 0, 1
            for sm in gsums:
                run_sum ^= F.mul(p, sm)
                p = F.mul(p, gstream[0])
            gok = True
            for r in range(gn_vars):
                c0, c1, c2 = gc[r]
                gok = gok and (c0 ^ (c0 ^ c1 ^ c2)) == run_sum
                z = gstream[1 + r]
                run_sum = c0 ^ F.mul(z, c1 ^ F.mul(z, c2))
            acc, p = 0, 1
            for i, j in gcomps:
                acc ^= F.mul(p, F.mul(gf[i], gf[j]))
                p = F.mul(p, gstream[0])
            gok = gok and acc == run_sum
        return {
            "workload": "one BivariateSumcheckProver, k=%d product claims over m=%d multilinears of 2^%d elements" % (gk, gm, gn_vars),
            "ms_per_prove": round(gms, 4), "elems_per_s": round(gm * gn / (gms * 1e-3), 1),
            "frac_of_64mN_at_8TBps": round(64.0 * gm * gn / (gms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4), "verifier_check": bool(gok),
            "per_prove": {"group_launches": (cnt["launches"] - c_before["launches"]) // 3, "claims_fused_with_their_folds": (cnt["jobs_fused"] - c_before["jobs_fused"]) // 3,
                          "rounds_on_the_host": (cnt["hosted_evals"] - c_before["hosted_evals"]) // 3, "plain_fold_launches": (cnt["prefolds"] + cnt["flushed_folds"] - c_before["prefolds"] - c_before["flushed_folds"]) // 3},
        }

    if rank == 0 and world == 1 and dist is None and not args.no_claim_groups:
        out["claim_groups"] = {}
        for name, gk, gn_vars in (("k4_m8", 4, min(24, n_global)), ("k50_m100_keccak_width", 50, min(22, n_global))):
            try:
                out["claim_groups"][name] = claim_groups_leg(gk, gn_vars)
            except Exception as ex:  # noqa: BLE001 -- a side measurement must not cost the headline line
                out["claim_groups"][name] = {"error": repr(ex)}

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
            mls = [oracle.random_b128(0xB1A50000 + j, 1 << fn) for j in range(m)]  # (the port folds in place)
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
    if peer is not None:
        peer.close()
    if shm is not None:
        shm.close()
    hal.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
