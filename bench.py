#!/usr/bin/env python3
"""bench.py -- scored frames/s of the template-scoring hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode fast|fast24|exact] [--replicas R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher environment starts its own N rank processes
(launch_ranks below: the parent never loads the HIP library or touches a GPU, nothing is re-exec'ed;
it relays rank 0's JSON line and exits with the worst child status).  Under torch.distributed.run the
ranks are the launcher's.

Workload (BASELINE.json configs[1], SURVEY.md section 8(d) "config 2"), per GPU:
1 000 synthetic utterances x 200 frames of binary edge features (45 x 8 = 360
per frame), scored against an 8-template bank (L = 32): one LAUNCH of the
scoring kernel.  One STEP = one pass over R device-resident copies of that corpus
(R launches back to back on one stream), R chosen so that a step lasts >= 5 ms and
its working set (R x 76 MB) is far larger than the 256 MiB Infinity Cache: every
launch streams its frames from HBM (SURVEY 8(d) "timing method").  The
cache-resident figure (one corpus scored over and over) is reported beside it.
With N GPUs every rank scores its own utterances (weak scaling; utterances are
independent, no data-path collective); the bank is generated on rank 0 and
broadcast once with RCCL before the timed region.

The product path is libtsr_hip.so through ctypes.  No torch anywhere: for N > 1 a
launcher (torch.distributed.run) only starts the processes; rendezvous, barrier,
max-over-ranks of the time and the hand-over of the RCCL unique id go over
tsr_hip.dist.HostGroup (standard-library sockets on the loopback interface).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))


def _wanted_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def launch_ranks(n, argv):
    """Parent of a self-launched N-GPU run: one child process per GPU with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would export), the same argv, a fresh shared
    secret for the host control plane.  This process imports neither tsr_hip nor torch and makes no GPU call.
    Rank 0's stdout is relayed line by line by a reader thread; the others' stdout goes to stderr.  The watchdog
    runs beside the relay from the start: once any rank has failed, the rest get TSR_LAUNCH_GRACE_S (20 s) to
    leave through the control plane's own errors and are then terminated by PID -- also when rank 0 sits inside a
    collective and never closes its stdout; TSR_LAUNCH_DEADLINE_S (3600 s) bounds the whole run the same way.
    Children are only ever started fresh; nothing is re-exec'ed."""
    import secrets
    import socket
    import subprocess
    import threading
    grace = float(os.environ.get("TSR_LAUNCH_GRACE_S", "20"))
    deadline = time.time() + float(os.environ.get("TSR_LAUNCH_DEADLINE_S", "3600"))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               TORCHELASTIC_RUN_ID="bench%d_%d" % (os.getpid(), int(time.time())), TSR_CTRL_SECRET=secrets.token_hex(32),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("TSR_CTRL_PORT", None)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def relay():
        for line in procs[0].stdout:      # rank 0 prints the one JSON line; ends when rank 0's stdout closes
            out = sys.stdout if line.lstrip().startswith("{") else sys.stderr   # (libraries' banners do not belong on stdout)
            out.write(line)
            out.flush()

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    gave_up = None
    try:
        failed_at = None
        while any(p.poll() is None for p in procs):
            now = time.time()
            if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
                failed_at = now
            if failed_at is not None and now - failed_at > grace:
                gave_up = "a rank failed and the others did not leave within %.0f s" % grace
                break
            if now > deadline:
                gave_up = "the run passed its deadline (TSR_LAUNCH_DEADLINE_S)"
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        reader.join(5)
    codes = [p.returncode for p in procs]
    if gave_up is not None:
        sys.stderr.write("bench.py launcher: %s; exit codes %s\n" % (gave_up, codes))
    worst = max((c for c in codes), key=lambda c: (c != 0, abs(c)))
    worst = worst if worst >= 0 else 128 - worst
    return worst if worst != 0 or gave_up is None else 1


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _wanted_gpus(sys.argv[1:]) > 1:
    raise SystemExit(launch_ranks(_wanted_gpus(sys.argv[1:]), sys.argv[1:]))   # before anything loads HIP

import tsr_hip  # noqa: E402  (loads libtsr_hip.so before anything else touches HIP)
from tsr_hip import dist as tdist  # noqa: E402
from tsr_hip import synth  # noqa: E402

CFG = dict(N=1000, T=200, F=45, E=8, M=8, L=32)
D = CFG["F"] * CFG["E"]
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
MFMA_I8_PEAK_TOPS = 5000.0    # dense int8 MFMA = 2x bf16 (~2.5 PF) per clock
MFMA_FP6_PEAK_TOPS = 10000.0  # dense fp6 / fp4 MFMA (block-scaled forms) = 4x bf16 per clock (MI355X_MICROARCH.md)
VALU_F32_PEAK_TFLOPS = 157.3
L3_BYTES = 256 << 20
ALG_BYTES_PER_FRAME = D + 6   # D u8 in + f32 max + 2-byte id
FLAGS = {"mx": tsr_hip.TSR_FAST | tsr_hip.TSR_MX, "fast": tsr_hip.TSR_FAST, "fast24": tsr_hip.TSR_FAST | tsr_hip.TSR_W24,
         "exact": tsr_hip.TSR_EXACT}
DTYPE = {"mx": "fp6 x fp4 (13-bit weights as 3 fp6 digits, binary frames as fp4; exact integer sums in one f32 accumulator)",
         "fast": "i8 (2 limbs, i32 accumulate)", "fast24": "i8 (3 limbs, i32 accumulate)", "exact": "f32"}
KERNEL_NAMES = {0: "none", 1: "exact_bank_kernel", 2: "exact_mfma_kernel", 3: "fast_bank_kernel / alt_bank_kernel", 4: "tap2_bank_kernel",
                5: "mx_bank_kernel"}
ALG_FLOPS = 2.0 * D * CFG["M"] * CFG["L"] * (CFG["T"] - CFG["L"] + 1) * CFG["N"]   # per launch (valid windows)


def _score_reference(ref, feats, offs, bank, n0, n1):
    cllf = ref[0]
    for n in range(n0, n1):
        E = feats[offs[n]:offs[n + 1]]
        row = cllf.detect(E, bank[0][0]) + np.float32(bank[0][1])
        for W, c in bank[1:]:
            row = np.maximum(cllf.detect(E, W) + np.float32(c), row)


def cpu_baseline(corpus, budget_s=10.0):
    """Single-core CPU baselines on a bounded sample of the same workload.  "port": oracle/'s C
    restatement (always travels).  "reference": the reference's own compiled Cython kernels when
    oracle/_ref travelled with the repo.  The headline `value` is the reference's when present, else
    the port's; both are listed.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_ref
    import oracle
    oracle.build()
    feats, offs, bank = corpus["feats"], corpus["offsets"], corpus["bank"]
    ref = build_ref.load()

    def timed(run):
        t0 = time.perf_counter()
        run(0, 8)
        per_utt = (time.perf_counter() - t0) / 8
        n = int(max(16, min(len(offs) - 1, budget_s / max(per_utt, 1e-6))))
        t0 = time.perf_counter()
        run(0, n)
        dt = time.perf_counter() - t0
        frames = int(offs[n] - offs[0])
        return {"value": frames / dt, "unit": "frames/s", "cores": 1,
                "sample": "%d of %d utterances (%d frames) of the same corpus, %.1f s, one thread"
                          % (n, len(offs) - 1, frames, dt)}

    port = timed(lambda a, b: oracle.score_corpus(feats[offs[a]:offs[b]], offs[a:b + 1] - offs[a], bank))
    out = dict(port, kind="port", port=dict(port))
    if ref is not None:
        r = timed(lambda a, b: _score_reference(ref, feats, offs, bank, a, b))
        out = dict(r, kind="reference", port=dict(port), reference=dict(r))
    return out


def _allcores_worker(args):
    """(child process, CPU only) score a slice of utterances with the compiled reference or the oracle."""
    seed, n0, n1 = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    c = synth.corpus(seed, **CFG)
    offs = c["offsets"]
    ref = None
    try:
        import build_ref
        ref = build_ref.load()
    except Exception:
        ref = None
    t0 = time.perf_counter()
    if ref is not None:
        _score_reference(ref, c["feats"], offs, c["bank"], n0, n1)
    else:
        oracle.score_corpus(c["feats"][offs[n0]:offs[n1]], offs[n0:n1 + 1] - offs[n0], c["bank"])
    return time.perf_counter() - t0, int(offs[n1] - offs[n0]), ref is not None


def cpu_allcores():
    """the single-process CPU path over utterances on every host core (one process per
    core, started BEFORE this process touches the GPU).  ~16 utterances per worker."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0))
    per = 16
    jobs = [(2, i * per, (i + 1) * per) for i in range(min(cores, CFG["N"] // per))]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        res = pool.map(_allcores_worker, jobs)
    wall = time.perf_counter() - t0
    frames = sum(r[1] for r in res)
    busy = max(r[0] for r in res)   # the slowest worker's scoring time (corpus generation excluded)
    return {"value": frames / busy, "unit": "frames/s", "cores": len(jobs), "kind": "reference" if all(r[2] for r in res) else "port",
            "sample": "%d utterances per core on %d cores, slowest worker %.1f s (pool wall %.1f s incl. data generation)"
                      % (per, len(jobs), busy, wall)}


def extras(ctx, corpus, bank, flags):
    """Secondary numbers: the host-buffer boundary including PCIe, and config 3."""
    frames = int(corpus["offsets"][-1])
    best = None
    for _ in range(3):  # upload + score + download through the host-pointer entry points
        ctx.sync()
        t0 = time.perf_counter()
        c = tsr_hip.Corpus(corpus["feats"], corpus["offsets"], ctx=ctx)
        c.score(bank, flags)
        c.download()
        dt = time.perf_counter() - t0
        c.close()
        best = dt if best is None else min(best, dt)
    res = {"pcie_inclusive": {"frames_per_s": frames / best, "ms": best * 1e3,
                              "what": "tsr_corpus_create (pageable host -> HBM) + score + tsr_corpus_download, best of 3"}}
    res["config3_byte_rows"] = config3(ctx, corpus, flags, layout=tsr_hip.TSR_PART_LAYOUT_BYTES)   # round 2's resident form
    return res


def config3(ctx, corpus, flags, layout=None):
    """BASELINE config 3 on the same corpus: raw edge maps -> code_parts(K=50, thr=12) -> argmax -> spread(2,2) ->
    8 x 32-tap bank on the D = 2050 part features, everything device-resident; the part corpus stays on the device as
    BIT rows (51 MB instead of 404 MB of bytes) and the bank pass reads those.  hipEvent times on the launch stream;
    the front-end call also allocates its output corpus (wall time beside it).  Never part of `value`."""
    frames = int(corpus["offsets"][-1])
    rng = np.random.default_rng(3)
    lp, lip = synth.parts(rng, 50)
    parts = tsr_hip.Parts(lp, lip, ctx=ctx)
    bank3 = tsr_hip.Bank(synth.bank(rng, synth.background(rng, (CFG["F"] - 4, 50)), [CFG["L"]] * CFG["M"]), ctx=ctx)
    edges = tsr_hip.Corpus(corpus["feats"], corpus["offsets"], ctx=ctx)
    old = ctx.get_option(tsr_hip.TSR_OPT_PART_LAYOUT)
    if layout is not None:
        ctx.set_option(tsr_hip.TSR_OPT_PART_LAYOUT, layout)
    fe_wall, fe_ms, bank_ms, lay = [], [], [], None
    try:
        for _ in range(5):
            ctx.sync()
            t0 = time.perf_counter()
            ctx.timer_start()
            pc = edges.part_features(parts, 12, spread=(2, 2))   # synchronous: returns the new corpus
            fe_ms.append(ctx.timer_stop())
            fe_wall.append((time.perf_counter() - t0) * 1e3)
            pc.score(bank3, flags)
            ctx.sync()
            ctx.timer_start()
            for _ in range(4):
                pc.score(bank3, flags)
            bank_ms.append(ctx.timer_stop() / 4)
            lay = pc.layout()
            pc.close()
    finally:
        ctx.set_option(tsr_hip.TSR_OPT_PART_LAYOUT, old)
    info = parts.info()
    edges.close()
    fe, bk = min(fe_ms[1:]), min(bank_ms[1:])
    traffic = None
    bank_kernel = KERNEL_NAMES.get(ctx.get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL), "?")
    # the stored traffic of the chain as this run's bank pass ran it (round 4: the mx kernel in feature phases; round 3: int8 K-phases)
    tn = "r04_config3_traffic.json" if bank_kernel == KERNEL_NAMES.get(tsr_hip.TSR_KERNEL_MX) else "r03_config3_traffic.json"
    tp = os.path.join(ROOT, "profiles", tn)
    if layout is None and os.path.exists(tp):
        tj = json.load(open(tp))
        traffic = {"hbm_bytes_per_chain": tj["hbm_bytes_per_chain"], "ratio_to_algorithmic": tj["ratio_to_algorithmic"],
                   "source": "profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" % tn,
                   "measured_in_this_run": False}
    return {"front_end_stream_ms": fe, "front_end_wall_ms": min(fe_wall[1:]), "bank_D2050_kernel_ms": bk,
            "frames_per_s": frames / ((fe + bk) * 1e-3), "frames": frames,
            "part_corpus": {"layout": "+".join(k for k in ("bits", "bytes") if lay[k]), "resident_MB": lay["feature_bytes"] / 1e6},
            "patch_locations": info["locations"], "f64_rechecked": info["rechecked"], "traffic": traffic,
            "what": "code_parts(K=50,thr=12) -> argmax -> spread(2,2) -> 8x32-tap bank on D=2050, one GPU; front_end_stream_ms = "
                    "hipEvents around tsr_corpus_part_features (coder, f64 re-evaluation, dilation and the allocation of the "
                    "output corpus), bank = one pass with the headline's flags over the resident part corpus (bank_kernel says which kernel served it)",
            "bank_kernel": bank_kernel}


def long_corpus(ctx, corpus, bank, flags, times=10):
    """The same kernel on `times` x the utterances in ONE launch (10 000 utterances of config 2's shape): the
    per-workgroup prologue (bank -> registers) and the launch itself are paid once per `times` corpora -- the asymptote
    of the launch-per-1000-utterances figure in `value`.  Never part of `value`."""
    feats = np.concatenate([corpus["feats"]] * times)
    offs = np.concatenate([corpus["offsets"][:-1] + k * int(corpus["offsets"][-1]) for k in range(times)] +
                          [np.array([times * int(corpus["offsets"][-1])], np.int64)])
    big = [tsr_hip.Corpus(feats, offs, ctx=ctx) for _ in range(6)]   # six copies (4.3 GB of byte rows, 0.6 GB of bit rows: past the 256 MiB Infinity Cache), scored in turn
    del feats
    time_launches(ctx, big, bank, flags, 4)             # (the first timed loops of a process run ~5 % slower than the later ones: clocks)
    ms, _ = time_launches(ctx, big, bank, flags, 8)
    frames = int(offs[-1])
    s0, _ = big[0].download()
    same = bool(np.array_equal(s0[:int(corpus["offsets"][-1])], s0[-int(corpus["offsets"][-1]):]))   # first and last copy of the corpus
    for c in big:
        c.close()
    return {"utterances_per_launch": (len(offs) - 1), "frames_per_launch": frames, "kernel_ms_per_launch": ms,
            "kernel_us_per_1000_utterances": ms * 1e3 / times, "frames_per_s": frames / (ms * 1e-3),
            "alg_tflops": ALG_FLOPS * times / (ms * 1e-3) / 1e12, "frac_of_int8_peak": ALG_FLOPS * times / (ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
            "frac_of_hbm_peak": ALG_BYTES_PER_FRAME * frames / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "hbm_gbs_alg": ALG_BYTES_PER_FRAME * frames / (ms * 1e-3) / 1e9, "first_and_last_copy_score_alike": same,
            "what": "config 2's corpus %d times over in one launch: prologue and launch amortised %dx" % (times, times)}


def two_streams(corpus, flags, device, copies=48, passes=12):
    """Independent corpora dealt round-robin to two contexts (= two HIP streams) of the device: workgroups of the next
    launch start on CUs as those of the running one retire, so the serial phases of a launch (staging, exchange, finish,
    the launch itself) overlap with the other stream's matrix work.  Whole-pipeline throughput by the wall clock, beside
    the same loop on one stream (tools/time_streams.py: 23.8 -> 21.2 us per launch; two rank processes on one GPU reach
    the same, profiles/r04_selflaunch_2rank_rehearsal.json); never part of `value` (the headline keeps one launch at a
    time, so that its kernel time is what rocprofv3 shows per kernel and `roofline.achieved` needs no concurrency factor)."""
    frames = int(corpus["offsets"][-1])
    ctxs = [tsr_hip.Context(device) for _ in range(2)]
    banks = [tsr_hip.Bank(corpus["bank"], ctx=x) for x in ctxs]
    reps = [[tsr_hip.Corpus(corpus["feats"], corpus["offsets"], ctx=x) for _ in range(copies)] for x in ctxs]

    def run(streams):
        for s in streams:
            for cp in reps[s]:
                cp.score(banks[s], flags)
        for x in ctxs:
            x.sync()
        t0 = time.perf_counter()
        for _ in range(passes):
            for i in range(copies):
                for s in streams:
                    reps[s][i].score(banks[s], flags)
        for x in ctxs:
            x.sync()
        return (time.perf_counter() - t0) / (passes * copies * len(streams))
    run([0]), run([0, 1])                               # (as above: the first loops are slower whatever they hold)
    one, two = run([0]), run([0, 1])
    a, b = reps[0][0].download(), reps[1][copies - 1].download()
    same = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))
    for r in reps:
        for cp in r:
            cp.close()
    for bk in banks:
        bk.close()
    for x in ctxs:
        x.close()
    return {"streams": 2, "copies_per_stream": copies, "us_per_launch_one_stream_wall": one * 1e6, "us_per_launch_two_streams_wall": two * 1e6,
            "frames_per_s": frames / two, "speedup_over_one_stream": one / two, "hbm_gbs_alg": ALG_BYTES_PER_FRAME * frames / two / 1e9,
            "frac_of_hbm_peak": ALG_BYTES_PER_FRAME * frames / two / 1e9 / HBM_PEAK_GBS, "first_and_last_copy_score_alike": same,
            "what": "config 2, every launch on its own device-resident copy, launches dealt to two streams (wall clock over %d launches)" % (2 * passes * copies)}


def time_launches(ctx, replicas, bank, flags, steps):
    """`steps` passes over the replica list -> (ms per launch by hipEvents on the launch stream, wall s)."""
    ctx.sync()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(steps):
        for c in replicas:
            c.score(bank, flags)
    ms = ctx.timer_stop()
    ctx.sync()
    return ms / (steps * len(replicas)), time.perf_counter() - t0


def parity_check(oracle, corpus, dev, seed, n=6):
    """A seeded random sample of utterances spread over the whole corpus, what was just computed on
    `dev` against the oracle: |a-b| <= 1e-4 max(|b|, 1) (1e-4 relative fp32, BASELINE north_star)."""
    offs = corpus["offsets"]
    scores, _ = dev.download()
    rng = np.random.default_rng(seed)
    worst = 0.0
    for u in sorted(rng.choice(len(offs) - 1, size=n, replace=False)):
        sl = slice(int(offs[u]), int(offs[u + 1]))
        so, _ = oracle.score_bank(corpus["feats"][sl], corpus["bank"])
        worst = max(worst, float(np.max(np.abs(scores[sl] - so) / np.maximum(np.abs(so), 1.0))))
    if worst > 1e-4:
        raise SystemExit("parity check failed in bench: rel err %.3g" % worst)
    return worst


_FAST_KERNEL_OPT = [0]   # the context's TSR_OPT_FAST_KERNEL (set in main once the context exists)


def roofline_of(mode, per_launch_s, frames, traffic):
    alg_bytes = ALG_BYTES_PER_FRAME * frames
    achieved = alg_bytes / per_launch_s / 1e9
    hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "alg_bytes_per_launch": alg_bytes}
    if mode == "exact":
        # VALU-bound by construction (sequential f32 adds in the reference's order); the HBM
        # roofline is what the contract asks to be reported against
        return dict(hbm, bound="hbm", traffic=traffic, kernel_ms_per_launch=per_launch_s * 1e3,
                    valu_tflops=ALG_FLOPS / per_launch_s / 1e12, valu_peak_tflops=VALU_F32_PEAK_TFLOPS)
    tops = ALG_FLOPS / per_launch_s / 1e12
    if mode == "mx":
        # fp6 x fp4 at ~10 PF: machine balance 1250 op/B against the path's 425 flop/B -- by algorithmic work HBM binds
        # (73.2 MB at 8 TB/s = 9.2 us against 31.15 Gflop at 10 PF = 3.1 us).  What the kernel issues is three digit
        # products per algorithmic MAC on 33 tap rows x 384 features for windows of 32 x 360 (x 170/169 start pairs):
        # 9.9 us of matrix work at the spec peak -- the two bounds are level, and both are reported.
        # K as the kernel lays it out (csrc/detect_mx.hip): L + 1 tap rows of D // 32 full chunks, the leftover features of a
        # row either a chunk of their own or (D % 32 <= 8) one chunk per four tap rows of a frame parity; K-blocks of four chunks
        rows_n, nfull, lo = CFG["L"] + 1, D // 32, D % 32
        if 1 <= lo <= 8:
            chunks = rows_n * nfull + sum(math.ceil(len(range(p, rows_n, 2)) / 4) for p in (0, 1))
        else:
            chunks = rows_n * (nfull + (1 if lo else 0))
        limbs, kpad, rows = 3, 128 * math.ceil(chunks / 4) / (rows_n * D), rows_n / CFG["L"]
        pairs = 2 * math.ceil((CFG["T"] - CFG["L"] + 1) / 2) / (CFG["T"] - CFG["L"] + 1)
        issued = tops * limbs * kpad * rows * pairs
        return dict(hbm, bound="hbm", traffic=traffic, kernel_ms_per_launch=per_launch_s * 1e3, alg_flops_per_launch=ALG_FLOPS,
                    mfma={"achieved": tops, "peak": MFMA_FP6_PEAK_TOPS, "unit": "TFLOP/s", "frac": tops / MFMA_FP6_PEAK_TOPS,
                          "issued": {"digit_macs_per_alg_mac": limbs, "k_pad": kpad, "tap_rows": rows, "start_pairs": pairs,
                                     "fp6_tops": issued, "frac_of_peak": issued / MFMA_FP6_PEAK_TOPS, "kernel": "mx_bank_kernel",
                                     "what": "fp6 x fp4 ops the kernel issues = algorithmic x 3 digits x K padding (%d K-elements for %d "
                                             "tap rows x %d features) x tap rows (%d/%d) x start pairs"
                                             % (128 * math.ceil(chunks / 4), rows_n, D, rows_n, CFG["L"])},
                          "peak_note": "dense fp6 / fp4 MFMA = 4x the ~2.5 PF bf16 dense peak (MI355X_MICROARCH.md)"},
                    bound_note="algorithmic intensity 425 flop/B is below the fp6 machine balance of 1250 op/B: HBM is the "
                               "binding roofline of this path at this precision; `achieved` = algorithmic bytes (366 B per "
                               "frame: D u8 in + score + id out) per launch / kernel time.  The kernel itself reads the "
                               "corpus' resident bit rows (48 B per frame), so its measured traffic is BELOW the algorithmic "
                               "bytes.  The matrix-core view of the same launch is under `mfma`.")
    # 425 flop/B at M=8 vs a machine balance of ~600 int8-op/B -- close to the
    # ridge, and each algorithmic MAC costs 2 (3) int8 limb-MACs, so the matrix cores bind
    limbs = 2 if mode == "fast" else 3
    # which two-limb kernel tsr_corpus_score picks (tsr_api.hip): the tap-pair kernel when the resident rows are an
    # odd number (<= 23) of 16-byte slots
    dp16 = math.ceil(D / 16)
    forced = _FAST_KERNEL_OPT[0]
    tap_pair = mode == "fast" and (forced == tsr_hip.TSR_FAST_KERNEL_TAP2 or
                                   (forced == tsr_hip.TSR_OPT_AUTO and dp16 % 2 == 1 and dp16 <= 23))
    if tap_pair:   # K = two resident rows of 16 * dp16 bytes per tap pair; odd L would add half a tap per template
        kpad = (16 * dp16) / D * (2 * math.ceil(CFG["L"] / 2)) / CFG["L"]
        kpad_what = "K padding (%d/%d; tap-pair kernel)" % (16 * dp16, D)
    else:          # K padded to whole 32-feature MFMA blocks
        kpad = (math.ceil(D / 32) * 32) / D
        kpad_what = "K padding (%d/%d)" % (math.ceil(D / 32) * 32, D)
    tail = CFG["T"] / (CFG["T"] - CFG["L"] + 1)               # tail frames run through the matrix cores too
    issued = tops * limbs * kpad * tail
    return {"bound": "mfma", "achieved": tops, "peak": MFMA_I8_PEAK_TOPS, "unit": "TFLOP/s",
            "frac": tops / MFMA_I8_PEAK_TOPS, "traffic": traffic,
            "kernel_ms_per_launch": per_launch_s * 1e3, "alg_flops_per_launch": ALG_FLOPS,
            "issued": {"limb_macs_per_alg_mac": limbs, "k_pad": kpad, "tail_rows": tail,
                       "int8_tops": issued, "frac_of_peak": issued / MFMA_I8_PEAK_TOPS,
                       "kernel": "tap2_bank_kernel" if tap_pair else ("fast_bank_kernel" if limbs == 2 else "alt_bank_kernel"),
                       "what": "int8 ops the kernel issues = algorithmic x limbs x %s x tail frames (%d/%d)"
                               % (kpad_what, CFG["T"], CFG["T"] - CFG["L"] + 1)},
            "peak_note": "dense int8 MFMA = 2x the ~2.5 PF bf16 dense peak (MI355X_MICROARCH.md)",
            "hbm": hbm}


def stored_traffic(mode):
    """HBM bytes per launch from the PMC passes committed under profiles/ (collected by tools/traffic.sh on
    the > L3 working set, gfx950 corrections applied) -- NOT measured in this run."""
    for name in ("r04_traffic_%s.json" % mode, "r03_traffic_%s.json" % mode, "r02_traffic_%s.json" % mode, "traffic_%s.json" % mode):  # newest round first
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            j = json.load(open(p))
            return j.get("hbm_bytes_per_launch"), {"source": "profiles/" + name, "measured_in_this_run": False,
                                                   "working_set": j.get("working_set", "L3-resident (round 1)")}
    return None, {"source": None, "measured_in_this_run": False}


def dry_run_ranks(world, rank):
    """The host side of an N-rank run with the GPU work left out (tests/test_bench_launch_cpu.py)."""
    if os.environ.get("TSR_DRY_RUN_SCENARIO") == "rank1_dies_rank0_hangs":
        # what a crashed rank looks like to the launcher while rank 0 sits inside a collective: no output, no exit
        if rank == 1:
            raise SystemExit(7)
        time.sleep(600)
    group = tdist.HostGroup.from_env(timeout=60) if world > 1 else tdist.HostGroup(0, 1)
    uid = tdist.exchange_unique_id(group, lambda: bytes(range(128)))
    bank = tdist.broadcast_object(group, synth.corpus(2, N=2, T=40, F=5, E=4, M=2, L=8)["bank"] if rank == 0 else None)
    group.barrier()
    wall, = tdist.max_over_ranks(group, 0.001 * (rank + 1))
    ranks = tdist._allgather(group, [rank, int(os.environ.get("LOCAL_RANK", "-1")), os.getpid()])
    if rank == 0:
        print(json.dumps({"metric": "scored frames/sec (whole node) + achieved HBM GB/s vs roofline", "dry_run": True,
                          "n_gpus": world, "value": None, "max_over_ranks_s": wall, "ranks": ranks,
                          "bank_templates": len(bank), "uid_ok": uid == bytes(range(128))}))
    group.barrier()
    group.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["auto", "mx", "fast", "fast24", "exact"],
                    default=os.environ.get("TSR_BENCH_MODE", "auto"),
                    help="mx: 13-bit weights as three fp6 digits on the block-scaled matrix instruction (Toeplitz form); "
                         "fast: 16-bit fixed-point weights (2 int8 limbs); fast24: 23-bit (3 limbs); exact: f32 in the "
                         "reference's summation order; auto: the fastest of these that passes the in-run parity check")
    ap.add_argument("--replicas", type=int, default=0,
                    help="device-resident copies of the corpus per step (0: enough for >= 5 ms per step "
                         "and a working set of >= 4x the Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the other precision modes (\"modes\" in the JSON)")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if RCCL fails, hand the bank over the host channel instead and still "
                         "exit 0 (the line then carries \"degraded\": true)")
    ap.add_argument("--extras", action="store_true",
                    help="N=1 only: add the PCIe-inclusive rate and BASELINE config 3 (parts front-end + "
                         "D=2050 bank) to the JSON line under \"extras\" (never part of `value`)")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="(CPU test of the launcher) every rank runs the N > 1 host control flow -- rendezvous, id "
                         "hand-over, bank over the host channel, barriers, max over ranks -- without a GPU call")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.dry_run_ranks:
        return dry_run_ranks(world, rank)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    if world > 1:
        # one node by contract: RCCL's bootstrap rendezvous over the loopback interface (a container's hostname or
        # outward interface need not resolve); the data of the one collective goes over xGMI / shared memory either way
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    allcores = cpu_allcores() if (args.extras and world == 1) else None   # forks: before any GPU call
    # (TSR_DEVICE pins every rank to one GPU: a rehearsal of the N > 1 control flow on a one-GPU box)
    ctx = tsr_hip.Context(int(os.environ.get("TSR_DEVICE", local_rank)))
    _FAST_KERNEL_OPT[0] = ctx.get_option(tsr_hip.TSR_OPT_FAST_KERNEL)
    group = tdist.HostGroup.from_env() if world > 1 else tdist.HostGroup(0, 1)

    # ---- data: every rank its own utterances, one bank for all ----------------------
    corpus = synth.corpus(2 + 1000 * rank, **CFG)
    bank_bcast, degraded = "none (single GPU)", False
    if world > 1:
        real_bank = corpus["bank"]
        if rank != 0:  # same geometry, weights to be overwritten by rank 0's
            corpus["bank"] = tdist.placeholder_bank([W.shape[0] for W, _ in corpus["bank"]],
                                                    corpus["bank"][0][0].shape[1:])

        def all_ok(flag):
            return tdist.max_over_ranks(group, 0.0 if flag else 1.0)[0] == 0.0

        err = ""
        # RCCL prints a version banner on stdout when a communicator is created: stdout carries the one JSON line and nothing
        # else, so file descriptor 1 points at stderr while the library initialises and broadcasts
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            ctx.dist_init(tdist.exchange_unique_id(group, tsr_hip.unique_id), world, rank)
        except tsr_hip.TsrError as e:
            err = str(e)
        rccl = all_ok(not err)
        bank = tsr_hip.Bank(corpus["bank"], ctx=ctx)
        if rccl:
            try:
                bank.broadcast(0)  # RCCL over xGMI; the only collective on this path
            except tsr_hip.TsrError as e:
                err = str(e)
            rccl = all_ok(not err)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
        if rccl:
            bank_bcast = "rccl"
        else:
            why = "RCCL failed%s" % ((": " + err[:160]) if err else " on another rank")
            if not args.allow_fallback:
                if rank == 0:
                    print(json.dumps({"error": "bank broadcast: " + why, "degraded": True}))
                group.barrier()
                raise SystemExit(3)
            # --allow-fallback: hand the bank over the host channel so that the scaling run still measures
            # the (collective-free) data path; the line says so
            corpus["bank"] = tdist.broadcast_object(group, real_bank)
            bank = tsr_hip.Bank(corpus["bank"], ctx=ctx)
            bank_bcast, degraded = "host-channel fallback (%s)" % why, True
        del real_bank
    else:
        bank = tsr_hip.Bank(corpus["bank"], ctx=ctx)
    frames = int(corpus["offsets"][-1])
    first = tsr_hip.Corpus(corpus["feats"], corpus["offsets"], ctx=ctx)
    if args.mode == "auto":  # the fastest path this build has for the bank: mx where its kernel serves it, else fast, else exact
        try:
            first.score(bank, FLAGS["mx"])
            ctx.sync()
            args.mode = "mx" if ctx.get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL) == tsr_hip.TSR_KERNEL_MX else "fast"
        except tsr_hip.TsrError as e:
            if e.status != 4:
                raise
            args.mode = "exact"
    flags = FLAGS[args.mode]

    # ---- cache-resident probe (one corpus over and over): sizes the replica list, reported beside --------
    for _ in range(5):
        first.score(bank, flags)
    resident_ms, _ = time_launches(ctx, [first], bank, flags, 30)

    def n_replicas(ms_per_launch):
        if args.replicas > 0:
            return args.replicas
        # bytes a launch streams per frame: the resident byte rows + tables + results, or (mx) the bit rows + results
        per_frame = (16 * math.ceil(D / 128) + 8 + 4) if args.mode == "mx" else (D + 8 + 12)
        r = max(int(math.ceil(5.0 / ms_per_launch)), int(math.ceil(4.0 * L3_BYTES / (frames * per_frame))))
        return min(r, 192)     # <= 16 GB of HBM

    R = int(tdist.max_over_ranks(group, float(n_replicas(resident_ms)))[0])
    replicas = [first] + [tsr_hip.Corpus(corpus["feats"], corpus["offsets"], ctx=ctx) for _ in range(R - 1)]

    def barrier():
        ctx.sync()
        group.barrier()

    # ---- the timed region: K steps of R launches, bracketed by barriers, MAX over ranks ---------------
    # (setup, untimed: the first pass over the replicas builds their derived forms -- host planning between the launches -- and
    # a process reaches its steady launch rate only after ~35 ms of back-to-back launches, tools/time_steps.py: 26.5 -> 23.7 us;
    # four passes here, so that the W warmup steps and the K timed ones all run in that state)
    for _ in range(4):
        for c in replicas:
            c.score(bank, flags)
    for _ in range(args.warmup):
        for c in replicas:
            c.score(bank, flags)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        for c in replicas:
            c.score(bank, flags)
    kernel_ms = ctx.timer_stop()          # hipEvents on the launch stream
    ctx.sync()
    wall = time.perf_counter() - t0       # this rank: common start (barrier above) -> its own last step done
    barrier()                             # closing bracket: every rank has finished
    # MAX over ranks of the per-rank intervals.  The closing barrier itself (a TCP round of the host
    # control plane against K x 5 ms of work) is not GPU time of any rank.
    wall, kernel_ms = tdist.max_over_ranks(group, wall, kernel_ms)

    out = None
    if rank == 0:
        # ---- parity on what was just computed: utterances spread over the corpus, first and last replica
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        oracle.build()
        err = max(parity_check(oracle, corpus, replicas[0], 11), parity_check(oracle, corpus, replicas[-1], 12))
        per_launch_s = kernel_ms * 1e-3 / (args.steps * R)
        traffic, traffic_note = stored_traffic(args.mode)
        roofline = roofline_of(args.mode, per_launch_s, frames, traffic)
        roofline["traffic_note"] = traffic_note
        out = {
            "metric": "scored frames/sec (whole node) + achieved HBM GB/s vs roofline",
            "value": frames * R * world * args.steps / wall,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[args.mode],
            "data": "synthetic",
            "config": {"workload": "config 2: %d utterances x %d frames x D=%d binary features, %d-template bank, "
                                   "L=%d, max-score + template id per frame = one launch; one step = %d launches "
                                   "over %d device-resident copies of the corpus (%.0f MB working set per GPU, "
                                   "streamed from HBM%s)"
                                   % (CFG["N"], CFG["T"], D, CFG["M"], CFG["L"], R, R, R * frames * (D + 20) / 1e6,
                                      "; the mx kernel reads each copy's bit rows, %.0f MB of them" % (R * frames * 16 * math.ceil(D / 128) / 1e6)
                                      if args.mode == "mx" else ""),
                       "mode": args.mode, "frames_per_launch": frames, "launches_per_step": R,
                       "frames_per_gpu_per_step": frames * R, "parallelism": "utterance-sharded x%d" % world,
                       "bank_broadcast": bank_bcast, "parity_rel_err_sample": err,
                       "kernel": KERNEL_NAMES.get(ctx.get_option(tsr_hip.TSR_INFO_LAST_BANK_KERNEL), "?")},
            "roofline": roofline,
            "cache_resident": {"kernel_ms_per_launch": resident_ms, "frames_per_s": frames / (resident_ms * 1e-3),
                               "what": "the same launch on ONE corpus over and over (served from the 256 MiB Infinity "
                                       "Cache): round 1's headline, kept for comparison"},
        }
        if degraded:
            out["degraded"] = True
        out["config"]["window_templates_per_s"] = (CFG["N"] * CFG["M"] * (CFG["T"] - CFG["L"] + 1)
                                                   * R * world * args.steps / wall)
    # ---- the other precision modes, same invocation (rank 0's GPU only; never `value`) ------------------
    if world == 1 and not args.no_modes:
        modes = {}
        for mode in ("mx", "fast", "fast24", "exact"):
            fl = FLAGS[mode]
            try:
                for _ in range(3):
                    first.score(bank, fl)
                res_ms, _ = time_launches(ctx, [first], bank, fl, 10)
            except tsr_hip.TsrError:
                continue
            Rm = min(R, n_replicas(res_ms))
            for c in replicas[:Rm]:
                c.score(bank, fl)
            steps_m = max(2, min(args.steps, int(math.ceil(60.0 / (max(res_ms, 1e-3) * Rm)))))   # ~60 ms of launches
            ms, _ = time_launches(ctx, replicas[:Rm], bank, fl, steps_m)
            perr = parity_check(oracle, corpus, replicas[Rm - 1], 13)
            r = roofline_of(mode, ms * 1e-3, frames, None)
            modes[mode] = {"dtype": DTYPE[mode], "kernel_ms_per_launch": ms, "frames_per_s": frames / (ms * 1e-3),
                           "launches_per_step": Rm, "cache_resident_kernel_ms_per_launch": res_ms,
                           "parity_rel_err_sample": perr,
                           "alg_tflops": ALG_FLOPS / (ms * 1e-3) / 1e12, "hbm_gbs_alg": r.get("hbm", r)["achieved"]}
        out["modes"] = modes
    if rank == 0:
        if world == 1 and not args.no_modes:
            for c in replicas[1:]:      # (the replica list has done its work: make room)
                c.close()
            out["configs"] = {"config3": config3(ctx, corpus, flags)}
            if args.mode in ("fast", "mx"):
                out["configs"]["config2_x10_one_launch"] = long_corpus(ctx, corpus, bank, flags)
                out["configs"]["config2_two_streams"] = two_streams(corpus, flags, int(os.environ.get("TSR_DEVICE", local_rank)))
        if world == 1 and args.extras:
            out["extras"] = extras(ctx, corpus, bank, flags)
            out["extras"]["cpu_allcores"] = allcores
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(corpus)
        print(json.dumps(out))
    group.barrier()
    if world > 1:
        ctx.dist_finalize()
    group.close()


if __name__ == "__main__":
    main()
