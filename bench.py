#!/usr/bin/env python3
"""bench.py — audio-seconds/sec of WavTokenizer encode_infer + decode on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--arch hop600|hop320] [--clips 64]

One "step" = one encode_infer + decode round trip of `--clips` independent 3 s / 24 kHz clips
per GPU (BASELINE.json configs[1]: WavTokenizer-small-600, batch 64 x 3 s), inputs already
resident in HBM, synthetic clips and random-init weights of the real architecture
(wavtokenizer_amd/synth.py).  For N > 1 the driver launches one rank per GPU with
torch.distributed.run; clips shard across ranks with no collective on the data path (weak
scaling) and the per-step codes all-gather + waveform gather to rank 0 over RCCL is inside the
timed region.  Rank 0 prints ONE JSON line.

Timing: W warm-up steps, then R blocks of exactly K steps, each block bracketed by a barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over ranks.  `ms_per_step` / `value` are
the MEDIAN block (a single 0.13 s block is a thin basis: the min / max over the blocks are on the line too).

Schedule of the headline (config.lanes): at N = 1 the K steps of a block are issued alternately on TWO HIP streams
(sharding.StepRunner(lanes=2): step i+1's encode_infer runs beside step i's decode and fills its launch gaps and last-round
tails; every step is still one full encode_infer + decode of the batch and all K complete inside the bracket; same kernels,
same bits).  The one-stream schedule is measured in the same run (config.one_lane) and is what the roofline launch is timed
in, so that a second stream never lengthens the launch being measured.  `--lanes 1` makes the one-stream schedule the
headline.  N > 1 keeps one lane (the end-of-step exchange is ordered against the persistent LSTM on one stream: DESIGN 6).

Extra objects on the line:
  roofline      — dominant kernel (ConvNeXt pwconv1 GEMM + GELU, 12 launches per step): achieved =
                  algorithmic 2*M*N*K per launch / mean launch duration of those launches during the timed
                  steps.  The duration is taken ON THE DEVICE (wt_plan_set_timing("@cnx.pwconv1"): every
                  workgroup of the launch reads the constant 100 MHz clock on entry and exit; earliest entry
                  to latest exit), not between HIP events: an event record is a packet of its own, costs
                  4-7 us between two launches and is counted into the bracket (94.6 us between events
                  against 88.7 us in the rocprofv3 trace of the same run; with the stamps 84.7 against 86.6,
                  the difference being the dispatch latency rocprof counts: profiles/r03_trace_vs_stamps.txt),
                  and 24 such packets per step also cost the step itself 2 %.  The kernel evaluates every
                  fp32-equivalent product with THREE v_mfma_f32_16x16x32_f16 (split-f16, gemm16s.hip), so
                  its MFMA roofline in algorithmic (fp32-equivalent) FLOP/s is the dense f16 peak / 3.
  cpu_baseline  — the oracle (oracle/cpu_ref.py, same ATen op sequence as the reference) timed on
                  this host's cores, rank 0 at N=1 only, on a bounded sample of the same workload
                  (BASELINE.md section 3 protocol: 3 warm-ups, median of >= 10 round trips; B = 1, 16, 64; 8 threads too).
  other_configs — N=1 only: BASELINE configs[2] (hop-320, 64 x 3 s), the per-GPU share of configs[4]
                  (hop-600, 32 x 30 s: p50 encode_infer latency, codes/s) and the reference's own usage
                  (infer.py:44-70: one clip at a time, a different length per file, bandwidth_id on the GPU).
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F16_MFMA_TFLOPS = 2516.6     # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz)
PEAK_F16X3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0   # fp32-equivalent products on the f16 pipe: 3 MFMAs each
SAMPLE_RATE = 24000
# algorithmic GFLOP per 3 s clip (2*MAC: convs, linears, LSTM, VQ, attention)
GFLOP_PER_CLIP = {"hop600": 20.852, "hop320": 38.910}
WORKLOAD = {"hop600": "WavTokenizer-small-600-24k-4096 (40 tok/s), encode_infer+decode round trip, 24 kHz clips",
            "hop320": "WavTokenizer-small-320-24k-4096 (75 tok/s), encode_infer+decode round trip, 24 kHz clips"}


def gflop_per_clip(arch_name, arch, clip_seconds):
    """SURVEY 8(d) FLOPs are for 3 s clips; everything is linear in the length except attention (4*L^2*dim per clip)."""
    L = arch.frames(clip_seconds * SAMPLE_RATE)
    L3 = arch.frames(3 * SAMPLE_RATE)
    return GFLOP_PER_CLIP[arch_name] * clip_seconds / 3.0 + 4.0 * arch.dim * (L * L - L3 * L3 * clip_seconds / 3.0) / 1e9


def cpu_baseline(arch_name, sd, clips_np, clip_seconds):
    """Oracle on the host cores, BASELINE.md section 3: fp32, inputs in RAM, 3 warm-ups then the median of >= 10 round
    trips per row.  torch's default thread count (= all cores) oversubscribes this small model, so a few thread counts
    are tried on a B = 16 batch first; rows: B = 1 (BASELINE configs[0]), 16 and 64 at the best count, B = 1 and 16 at
    8 threads (comparable with the 8-CPU development container).  Bounded to ~2.5 minutes of CPU work."""
    from oracle.cpu_ref import OracleWavTokenizer
    from wavtokenizer_amd import NAMED_ARCHS
    orc = OracleWavTokenizer(NAMED_ARCHS[arch_name], sd)
    bw = torch.tensor([0])
    ncpu = os.cpu_count() or 1
    deadline = time.time() + 150.0

    def cpu_model():
        try:
            with open("/proc/cpuinfo") as f:
                for ln in f:
                    if ln.startswith("model name"):
                        return ln.split(":", 1)[1].strip()
        except OSError:
            pass
        import platform
        return platform.processor() or "unknown"

    spread = {}

    def rate(B, reps, warm=3):
        x = torch.from_numpy(clips_np[:B])
        ts = []
        with torch.inference_mode():
            for _ in range(warm):
                f, _ = orc.encode_infer(x, bw)
                orc.decode(f, bw)
                if time.time() > deadline:
                    break
            for _ in range(reps):
                t0 = time.perf_counter()
                f, _ = orc.encode_infer(x, bw)
                orc.decode(f, bw)
                ts.append(time.perf_counter() - t0)
                if time.time() > deadline and len(ts) >= 3:
                    break
        q = sorted(ts)
        spread[(B, torch.get_num_threads())] = {"p10_ms": round(1e3 * q[int(0.1 * (len(q) - 1))], 2), "p50_ms": round(1e3 * statistics.median(q), 2),
                                                "p90_ms": round(1e3 * q[int(round(0.9 * (len(q) - 1)))], 2)}
        return B * clip_seconds / statistics.median(ts), len(ts)

    tried = {}
    for nt in sorted({min(ncpu, n) for n in (8, 16, 32, 64)}):     # all-cores (>64) only thrashes
        torch.set_num_threads(nt)
        tried[nt] = rate(16, 2, warm=1)[0]
    best_nt = max(tried, key=tried.get)
    rows = {}
    torch.set_num_threads(best_nt)
    for B in sorted({1, min(16, len(clips_np)), min(64, len(clips_np))}):
        r, n = rate(B, 10)
        rows[f"B={B},threads={best_nt}"] = {"audio_s_per_s": round(r, 2), "round_trips": n, **spread[(B, best_nt)]}
    if best_nt != min(8, ncpu):
        torch.set_num_threads(min(8, ncpu))
        for B in sorted({1, min(16, len(clips_np))}):
            r, n = rate(B, 10)
            rows[f"B={B},threads={min(8, ncpu)}"] = {"audio_s_per_s": round(r, 2), "round_trips": n, **spread[(B, min(8, ncpu))]}
    torch.set_num_threads(best_nt)
    value = max(v["audio_s_per_s"] for k, v in rows.items() if k.endswith(f"threads={best_nt}"))
    return {"value": round(value, 2), "unit": "audio-s/s", "cores": best_nt, "kind": "port", "cpu_model": cpu_model(),
            "logical_cpus": ncpu, "rows": rows,
            "sample": "oracle/cpu_ref.py (the reference's ATen op sequence, fp32), encode_infer+decode round trips of %d s clips: "
                      "3 warm-ups then the median of up to 10 per row; thread sweep on B=16 (2 reps): %s; value = the fastest row at "
                      "%d threads; host has %d logical CPUs"
                      % (clip_seconds, ", ".join(f"{k}t={v:.1f}" for k, v in sorted(tried.items())), best_nt, ncpu)}


def pmc_traffic(arch_name, B, clip_seconds):
    """Bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (separate
    FETCH_SIZE / WRITE_SIZE passes, gfx950-corrected: tools/summarize_profile.py); None if there is
    no summary for this workload.  PMC counters cannot be read from inside the timed process."""
    import glob
    if arch_name != "hop600" or B != 64 or clip_seconds != 3:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        kernels = json.load(f)["kernels"]
    for name, v in kernels.items():
        if "gemm16s_kernel<128, 192, 4, 2, 3, 2, 1" in name:     # EPI_BIAS_GELU, S32 out (then the DBG and MFMA-shape parameters)
            return v["traffic_bytes_per_launch"]
    return None


def make_model(arch_name, dev):
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth
    arch = NAMED_ARCHS[arch_name]
    sd = synth.make_state_dict(arch, seed=0)
    t0 = time.perf_counter()
    model = WavTokenizer.from_arch(arch)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.eval().to(dev)
    model._ensure_engine()              # fold + pack + upload now, not inside the first step
    torch.cuda.synchronize()
    return model, arch, sd, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; the line carries the median block")
    ap.add_argument("--arch", default="hop600", choices=["hop600", "hop320"])
    ap.add_argument("--clips", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--clip-seconds", type=int, default=3, help="clip length (BASELINE configs[4] uses 30 s clips)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--lanes", type=int, default=2, help="N = 1 only: steps in flight on separate HIP streams (StepRunner(lanes=...)); "
                    "default 2 (config.lanes says so); the one-stream schedule is always measured too (config.one_lane) and "
                    "carries the roofline launch timing; --lanes 1 makes it the headline")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1: create a world-1 RCCL process group and run the end-of-step exchange (all_gather_into_tensor + "
                         "gather, async) inside the timed steps anyway: exercises library load, stream semantics and the "
                         "interplay with the persistent LSTM on the one GPU a development box has")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured path); gloo only rehearses the N>1 flow on a box with "
                         "fewer GPUs than ranks (ranks share GPUs, collectives go through host memory)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        args.gpus = world
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs: RCCL needs one GPU per rank (use --backend gloo to rehearse)")
    dev = torch.device("cuda", local_rank % max(1, ndev))
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.force_collectives:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from wavtokenizer_amd import synth, _capi
    from wavtokenizer_amd.sharding import StepRunner as Runner

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run_blocks(runner, steps, warmup, repeats, time_plan=None):
        """W warm-up steps, then `repeats` blocks of exactly `steps` steps; returns the per-block seconds (MAX over
        ranks) and, with time_plan = (plan, step-name filter), the HIP-event time of those launches."""
        for _ in range(warmup):
            runner.step()
        runner.drain()
        barrier()
        if time_plan:
            _capi.check(_capi.lib.wt_plan_set_timing(time_plan[0], time_plan[1]), "wt_plan_set_timing")
        blocks = []
        runner.own_blocks = []
        runner.wait_s, runner.exchanges = 0.0, 0
        for _ in range(repeats):
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                runner.step()
            runner.drain()
            barrier()
            dt = time.perf_counter() - t0
            runner.own_blocks.append(dt)                    # this rank's own clock; the line carries the MAX over ranks
            blocks.append(max_over_ranks(dt))
        kern = None
        if time_plan:
            tot_ms, n_l = ctypes.c_double(), ctypes.c_int64()
            _capi.check(_capi.lib.wt_plan_read_timing(time_plan[0], ctypes.byref(tot_ms), ctypes.byref(n_l), 1), "wt_plan_read_timing")
            _capi.lib.wt_plan_set_timing(time_plan[0], b"")
            kern = (tot_ms.value, n_l.value)
        return blocks, kern

    # ------------------------------------------------------------------------------ the headline workload
    model, arch, sd, load_s = make_model(args.arch, dev)
    clip_s = args.clip_seconds
    B, T = args.clips, clip_s * SAMPLE_RATE
    clips_np = synth.make_clips(B, T, seed=1000 * 2 + rank)       # seed = 1000*config + index
    wav = torch.from_numpy(clips_np).to(dev)
    bw = torch.tensor([0])
    L = arch.frames(T)
    forced = args.force_collectives and world == 1
    lanes = args.lanes if (world == 1 and not forced) else 1
    # one stream first: the schedule the roofline launch is timed in (and the headline with --lanes 1 / N > 1)
    runner = Runner(model, wav, bw, dist, world, rank, not args.no_gather, args.backend, lanes=1, force_collectives=forced)
    runner.step()                                                  # creates the plans
    runner.drain()
    dplan = next(p for k, (p, _w) in model._engine.plans.items() if k == (_capi.WT_PLAN_DECODE, B, L, model._graph_flags(B)))
    blocks, kern = run_blocks(runner, args.steps, max(0, args.warmup - 1), args.repeats, (dplan, b"cnx.pwconv1" if os.environ.get("WT_BENCH_TIMING") == "events" else b"@cnx.pwconv1"))     # (events: tools/trace_vs_events.sh only)
    model.check_status()
    one_lane_blocks = sorted(1e3 * b / args.steps for b in blocks)
    if lanes > 1:
        runner = Runner(model, wav, bw, None, 1, 0, False, args.backend, lanes=lanes)
        for _ in range(lanes):
            runner.step()
        runner.drain()
        blocks, _ = run_blocks(runner, args.steps, max(lanes, args.warmup), args.repeats)
        model.check_status()
    own_blocks = list(runner.own_blocks)

    # N > 1: what every rank saw, so that the first run on a real multi-GPU node explains itself (rank 0 prints it)
    ranks_info = None
    if world > 1:
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(dev), "device_index": dev.index,
                "world_size_seen": dist.get_world_size(), "backend": dist.get_backend(),
                "ms_per_step_own_clock": [round(1e3 * t / args.steps, 3) for t in own_blocks],
                "exchange_wait_ms_per_step": round(1e3 * runner.wait_s / max(1, runner.exchanges), 3),
                "exchanges": runner.exchanges,
                # (a lost co-residency would have raised in check_status() above)
                "persistent_lstm": bool(model.persistent_lstm)}      # the library's own word (a lost co-residency clears it)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_info = gathered

    p50_encode_ms = None
    if clip_s >= 30:       # BASELINE configs[4] also asks for the p50 latency of one encode_infer call
        lat = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model.encode_infer(wav, bandwidth_id=bw)
            e1.record()
            e1.synchronize()
            lat.append(e0.elapsed_time(e1))
        p50_encode_ms = statistics.median(lat)

    line = None
    if rank == 0:
        per_step = sorted(1e3 * b / args.steps for b in blocks)
        ms_per_step = statistics.median(per_step)
        value = world * B * clip_s / (ms_per_step * 1e-3)
        Mrows = B * L
        flops = 2.0 * Mrows * arch.intermediate_dim * arch.dim
        kern_ms = kern[0] / max(1, kern[1])
        achieved = flops / (kern_ms * 1e-3) / 1e12
        e2e_tflops = gflop_per_clip(args.arch, arch, clip_s) * B * 1e9 / (ms_per_step * 1e-3) / 1e12
        line = {
            "metric": "audio-seconds/sec encode+decode, 24 kHz %d s clips" % clip_s,
            "value": round(value, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (split-f16 x3 products)",
            "dtype_note": "fp32 accumulation everywhere; residual streams, norm / LSTM-cell arithmetic and outputs are fp32; every operand of a "
                          "dense layer is STORED between layers as a split pair of f16 numbers (S32: x = hi + lo * 2^-11, 22 significant "
                          "bits, f16 exponent range guarded by a status word) and each product is formed by 3 f16 MFMAs; the same workload "
                          "on the plain fp32 MFMA chain is other_configs.fp32_gemm_chain_64x3s; measured error of both against a float64 "
                          "run of the oracle: tests/test_gpu_parity.py::test_precision_against_float64 (profiles/r03_gpu_parity_summary.json)",
            "data": "synthetic",
            "timing": {"blocks": len(per_step), "steps_per_block": args.steps, "statistic": "median block",
                       "ms_per_step_min": round(per_step[0], 3), "ms_per_step_max": round(per_step[-1], 3),
                       "ms_per_step_blocks": [round(x, 3) for x in per_step],
                       "value_min": round(world * B * clip_s / (per_step[-1] * 1e-3), 1),
                       "value_max": round(world * B * clip_s / (per_step[0] * 1e-3), 1)},
            "config": {"workload": WORKLOAD[args.arch], "arch": args.arch, "clips_per_gpu": B,
                       "global_clips": world * B, "clip_seconds": clip_s, "frames_per_clip": L,
                       "codes_per_sec": round(world * B * L / (ms_per_step * 1e-3), 1),
                       **({"p50_encode_infer_ms_rank0": round(p50_encode_ms, 3)} if p50_encode_ms is not None else {}),
                       "weights": "random-init (synth seed 0)", "model_load_s": round(load_s, 2),
                       "hbm_weight_bytes": int(_capi.lib.wt_model_weight_bytes(model._engine.model)),
                       "parallelism": f"clips sharded dp{world}", "lanes": runner.lanes,
                       "schedule": ("the K steps of a block are issued alternately on %d HIP streams (step i+1's encode_infer beside step i's decode); "
                                    "each step is one full encode_infer + decode of the batch; all K complete inside the timed bracket" % runner.lanes)
                                   if runner.lanes > 1 else "one step after the other on one HIP stream",
                       "one_lane": {"ms_per_step": round(statistics.median(one_lane_blocks), 3), "ms_per_step_blocks": [round(x, 3) for x in one_lane_blocks],
                                    "audio_s_per_s": round(world * B * clip_s / (statistics.median(one_lane_blocks) * 1e-3), 1),
                                    "what": "the same workload, one step after the other on one stream (the r01-r03 headline schedule); "
                                            "the roofline launch is timed in this block"},
                       "persistent_lstm": bool(model.persistent_lstm),
                       **({"forced_collectives": "world-1 RCCL process group: codes all_gather_into_tensor + waveform gather (async_op) "
                                                 "inside every timed step"} if forced else {}),
                       **({"ranks": ranks_info} if ranks_info is not None else {}),
                       "gather": ("codes all_gather + waveform gather to rank 0 (%s), asynchronous: step i's exchange runs beside step i+1's decode "
                                  "(never beside the persistent LSTM), all finished inside the timed region" % ("RCCL" if args.backend == "nccl" else "gloo rehearsal"))
                       if world > 1 and not args.no_gather else "none"},
            "roofline": {"bound": "mfma",
                         "kernel": "wt::gemm16s_kernel<128,192,4,2,3,EPI_BIAS_GELU=2,OUT_S32=1> (ConvNeXt pwconv1 GEMM %dx%dx%d + GELU, "
                                   "split-f16: 3 x v_mfma_f32_16x16x32_f16 per fp32-equivalent product)" % (Mrows, arch.intermediate_dim, arch.dim),
                         "achieved": round(achieved, 2), "peak": round(PEAK_F16X3_TFLOPS, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F16X3_TFLOPS, 4), "traffic": pmc_traffic(args.arch, B, clip_s),
                         "peak_note": "algorithmic fp32-equivalent FLOP/s; peak = %.1f TF dense f16 MFMA / 3 MFMAs per product; the kernel "
                                      "issues %.0f TF/s of f16 MFMA work" % (PEAK_F16_MFMA_TFLOPS, 3 * achieved),
                         "algorithmic_bytes_per_launch": 4 * (Mrows * arch.dim + arch.intermediate_dim * arch.dim + Mrows * arch.intermediate_dim),
                         "flops_per_launch": flops, "avg_launch_ms": round(kern_ms, 4), "launches_timed": kern[1],
                         "launch_timing": "device clock (s_memrealtime, 100 MHz) of the launch's first workgroup entry to its last exit, every "
                                          "pwconv1 launch of the timed ONE-LANE steps (config.one_lane); no events in the stream "
                                          "(profiles/r03_trace_vs_stamps.txt compares it with the rocprofv3 trace and with bracketing HIP events in one run)",
                         "end_to_end_tflops": round(e2e_tflops, 2),
                         "end_to_end_frac": round(e2e_tflops / PEAK_F16X3_TFLOPS, 4)},
        }

    # ------------------------------------------------------------------------------ other configs (N = 1)
    if world == 1 and rank == 0 and not args.no_other_configs and args.arch == "hop600" and clip_s == 3:
        other = {}
        # the reference's own usage (infer.py:44-70): one clip per call, a different length per file, bandwidth_id built
        # on the GPU once; >= 20 distinct lengths between 1 and 10 s
        import numpy as np
        rng = np.random.default_rng(2024)
        lengths = sorted({int(x) for x in rng.integers(SAMPLE_RATE, 10 * SAMPLE_RATE, size=24)})
        bw_dev = torch.tensor([0]).to(dev)
        clips1 = [torch.from_numpy(synth.make_clips(1, t, seed=77 + i)).to(dev) for i, t in enumerate(lengths)]
        torch.cuda.synchronize()
        first, plan_ms = [], []
        for x in clips1:
            t0 = time.perf_counter()
            p = ctypes.c_void_p()
            _capi.check(_capi.lib.wt_plan_create(model._engine.model, _capi.WT_PLAN_ENCODE, 1, x.shape[1], 0, ctypes.byref(p)), "plan")
            plan_ms.append(1e3 * (time.perf_counter() - t0))
            _capi.lib.wt_plan_destroy(p)
            t0 = time.perf_counter()
            f, c = model.encode_infer(x, bandwidth_id=bw_dev)
            y = model.decode(f, bandwidth_id=bw_dev)
            torch.cuda.synchronize()
            first.append(1e3 * (time.perf_counter() - t0))
        steady = []
        for rep in range(3):
            for x in clips1:
                t0 = time.perf_counter()
                f, c = model.encode_infer(x, bandwidth_id=bw_dev)
                y = model.decode(f, bandwidth_id=bw_dev)
                torch.cuda.synchronize()
                if rep:
                    steady.append(1e3 * (time.perf_counter() - t0))
        # ... and back to back without a host synchronisation per file (what a pipelined caller gets)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(2):
            for x in clips1:
                f, c = model.encode_infer(x, bandwidth_id=bw_dev)
                y = model.decode(f, bandwidth_id=bw_dev)
        torch.cuda.synchronize()
        pipelined_ms = 1e3 * (time.perf_counter() - t0) / (2 * len(clips1))
        audio_s = sum(lengths) / SAMPLE_RATE
        other["infer_py_loop_B1_variable_length"] = {
            "reference_usage": "infer.py:44-70: one clip per encode_infer/decode call, a new length per file, bandwidth_id tensor on the GPU",
            "files": len(lengths), "seconds_per_file_min_max": [round(lengths[0] / SAMPLE_RATE, 2), round(lengths[-1] / SAMPLE_RATE, 2)],
            "plan_create_ms_median": round(statistics.median(plan_ms), 3),
            "first_call_ms_median": round(statistics.median(first), 3), "first_call_ms_max": round(max(first), 3),
            "steady_state_ms_median": round(statistics.median(steady), 3), "steady_state_ms_max": round(max(steady), 3),
            "pipelined_ms_per_file": round(pipelined_ms, 3),
            "audio_s_per_s_synchronised": round(audio_s / (sum(steady) / 2 / 1e3), 1),
            "audio_s_per_s_pipelined": round(audio_s / (pipelined_ms * len(lengths) / 1e3), 1),
            "plans_cached": len(model._engine.plans)}
        model.check_status()

        # the headline workload on the plain fp32 MFMA chain (v_mfma_f32_32x32x2_f32 everywhere): same arithmetic type end to
        # end, so the cost of fp32 products on the fp32 pipe is on the line beside the split-f16 path
        model.set_gemm_precision("f32")
        r32 = Runner(model, wav, bw, None, 1, 0, False, args.backend)
        r32.step()
        blk, _ = run_blocks(r32, 5, 1, 3)
        ms32 = statistics.median(1e3 * t / 5 for t in blk)
        model.check_status()
        model.set_gemm_precision("f16x3")
        model._engine.drop(lambda k: k[3] & _capi.WT_PLAN_FLAG_FP32_GEMM)
        other["fp32_gemm_chain_64x3s"] = {
            "what": "configs[1] with WT_PLAN_FLAG_FP32_GEMM: every dense layer on v_mfma_f32_32x32x2_f32 (fp32 operands in HBM, exact fp32 FMA chain)",
            "ms_per_step": round(ms32, 3), "audio_s_per_s": round(B * clip_s / (ms32 * 1e-3), 1),
            "end_to_end_tflops": round(gflop_per_clip(args.arch, arch, clip_s) * B * 1e9 / (ms32 * 1e-3) / 1e12, 2),
            "end_to_end_frac_of_fp32_mfma_peak_157.3": round(gflop_per_clip(args.arch, arch, clip_s) * B * 1e9 / (ms32 * 1e-3) / 1e12 / 157.3, 4),
            "steps": 5, "blocks": 3}
        del r32

        # the host-resident pipeline ("H2D reported separately"; the reference's script reads and writes files:
        # infer.py:44-70): pinned host buffers -> H2D on a copy stream -> encode_infer + decode + PCM16 on two lanes -> D2H of
        # the int16 samples on a second copy stream, double-buffered (sharding.HostPipeline)
        try:
            from wavtokenizer_amd.sharding import HostPipeline
            hp = HostPipeline(model, B, T, bw, lanes=2)
            for k in range(hp.lanes):
                hp.h_in[k].copy_(torch.from_numpy(clips_np))
            for _ in range(4):
                hp.step()
            hp.drain()
            blk = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    hp.step()
                hp.drain()
                blk.append(1e3 * (time.perf_counter() - t0) / args.steps)
            def busy():
                f_, _c = model.encode_infer(wav, bandwidth_id=bw)
                model.decode(f_, bandwidth_id=bw)
            h2d_ms, d2h_ms = hp.copy_times_ms(8, busy)     # the copy legs on a side stream beside a resident-input round trip
            torch.cuda.synchronize()
            model.check_status()
            ms_hp = statistics.median(blk)
            resident = line["ms_per_step"]
            pcm = hp.h_out[0].numpy()
            other["host_pipeline_64x3s"] = {
                "what": "configs[1] from and to HOST memory (infer.py:44-70), two lanes, each an in-order HIP stream: H2D of pinned fp32 "
                        "waveforms -> encode_infer + decode + PCM16 (wt_pcm16) -> D2H of the int16 samples; one lane's copies run beside "
                        "the other's kernels; h2d / d2h: the same copies timed on a side stream while the lanes compute",
                "ms_per_step": round(ms_hp, 3), "audio_s_per_s": round(B * clip_s / (ms_hp * 1e-3), 1), "ms_per_step_blocks": [round(x, 3) for x in sorted(blk)],
                "h2d_ms_per_step": round(h2d_ms, 3), "h2d_bytes": int(B * T * 4), "h2d_GBps": round(B * T * 4 / (h2d_ms * 1e-3) / 1e9, 1),
                "d2h_ms_per_step": round(d2h_ms, 3), "d2h_bytes": int(pcm.size * 2), "d2h_GBps": round(pcm.size * 2 / (d2h_ms * 1e-3) / 1e9, 1),
                "fraction_of_resident_rate": round(resident / ms_hp, 4),
                "bound": "compute (a lane's copies run beside the other lane's kernels)" if ms_hp < 1.1 * resident else
                         "the step is longer than the resident-input step by more than 10 %: see the copy times",
                "pcm16_nonzero": bool((pcm != 0).any()), "steps": args.steps, "blocks": 3}
            del hp
        except Exception as e:                      # a measurement row must not cost the headline line
            other["host_pipeline_64x3s"] = {"error": repr(e)[:300]}
        model._engine.drop(lambda k: len(k) >= 5 and k[4])

        # RCCL on the one GPU there is: a world-1 nccl process group with device_id, the exchange of
        # StepRunner(gather=True) - codes all_gather_into_tensor + waveform gather, async_op - for 20 steps beside the
        # persistent LSTM; the exchange order is asserted on the REAL codec's log and the gathered tensors against the step's own
        try:
            import torch.distributed as tdist
            from wavtokenizer_amd.sharding import check_exchange_order
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            own_pg = not tdist.is_initialized()
            if own_pg:
                tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            log = []
            rr = Runner(model, wav, bw, tdist, 1, 0, True, "nccl", log=log, force_collectives=True)
            n_st = 20
            ok = True
            prev = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_st):
                codes_i, out_i, res = rr.step()
                if res is not None and prev is not None:
                    ok = ok and torch.equal(res[0], prev[0]) and torch.equal(res[1], prev[1])
                prev = (codes_i, out_i)
            res = rr.drain()
            ok = ok and res is not None and torch.equal(res[0], prev[0]) and torch.equal(res[1], prev[1])
            torch.cuda.synchronize()
            ms_rc = 1e3 * (time.perf_counter() - t0) / n_st
            check_exchange_order(log, n_st)
            model.check_status()                    # an RCCL kernel in flight must not have tripped WT_STATUS_LSTM
            other["rccl_world1"] = {
                "what": "world-1 RCCL process group (device_id set): StepRunner(gather=True, force_collectives=True), 20 steps; codes "
                        "all_gather_into_tensor + waveform gather(async_op=True) issued behind step i+1's encode_infer, collected before "
                        "the step returns",
                "backend": tdist.get_backend(), "world_size": tdist.get_world_size(), "steps": n_st, "ms_per_step": round(ms_rc, 3),
                "exchanges": rr.exchanges, "exchange_wait_ms_per_step": round(1e3 * rr.wait_s / max(1, rr.exchanges), 3),
                "exchange_order_ok": True, "gathered_equals_own": bool(ok), "persistent_lstm_after": bool(model.persistent_lstm),
                "ranks": [{"rank": 0, "device": torch.cuda.get_device_name(dev), "device_index": dev.index}]}
            if own_pg:
                tdist.destroy_process_group()
        except Exception as e:
            other["rccl_world1"] = {"error": repr(e)[:300]}

        # BASELINE configs[4], per-GPU share: hop-600, 32 clips x 30 s
        wav30 = torch.from_numpy(synth.make_clips(32, 30 * SAMPLE_RATE, seed=1000 * 4)).to(dev)
        r30 = Runner(model, wav30, bw, None, 1, 0, False, args.backend)
        r30.step()
        blk, _ = run_blocks(r30, 5, 1, 3)
        ms30 = statistics.median(1e3 * b / 5 for b in blk)
        lat = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model.encode_infer(wav30, bandwidth_id=bw)
            e1.record()
            e1.synchronize()
            lat.append(e0.elapsed_time(e1))
        L30 = arch.frames(30 * SAMPLE_RATE)
        r30b = Runner(model, wav30, bw, None, 1, 0, False, args.backend, lanes=2)      # the headline's schedule on this shape
        r30b.step(); r30b.step(); r30b.drain()
        blk2, _ = run_blocks(r30b, 6, 2, 3)
        ms30_2 = statistics.median(1e3 * b / 6 for b in blk2)
        del r30b
        other["hop600_32x30s"] = {
            "baseline_config": "configs[4] per-GPU share: WavTokenizer-large-600 architecture, 32 clips x 30 s",
            "ms_per_step": round(ms30, 3), "audio_s_per_s": round(32 * 30 / (ms30 * 1e-3), 1),
            "ms_per_step_two_lanes": round(ms30_2, 3), "audio_s_per_s_two_lanes": round(32 * 30 / (ms30_2 * 1e-3), 1),
            "codes_per_sec": round(32 * L30 / (ms30 * 1e-3), 1), "p50_encode_infer_ms": round(statistics.median(lat), 3),
            "end_to_end_tflops": round(gflop_per_clip("hop600", arch, 30) * 32 * 1e9 / (ms30 * 1e-3) / 1e12, 2),
            "steps": 5, "blocks": 3}
        model.check_status()
        del wav30, r30
        model._engine.close()
        model._dirty = True
        torch.cuda.empty_cache()

        # BASELINE configs[2]: hop-320, 64 x 3 s
        m320, a320, _sd320, load320 = make_model("hop320", dev)
        wav320 = torch.from_numpy(synth.make_clips(64, 3 * SAMPLE_RATE, seed=1000 * 3)).to(dev)
        r320 = Runner(m320, wav320, bw, None, 1, 0, False, args.backend)
        r320.step()
        blk, _ = run_blocks(r320, 10, 2, 3)
        ms320 = statistics.median(1e3 * b / 10 for b in blk)
        other["hop320_64x3s"] = {
            "baseline_config": "configs[2]: WavTokenizer-small-320 (75 tok/s), 64 clips x 3 s",
            "ms_per_step": round(ms320, 3), "audio_s_per_s": round(64 * 3 / (ms320 * 1e-3), 1),
            "codes_per_sec": round(64 * a320.frames(3 * SAMPLE_RATE) / (ms320 * 1e-3), 1),
            "end_to_end_tflops": round(GFLOP_PER_CLIP["hop320"] * 64 * 1e9 / (ms320 * 1e-3) / 1e12, 2),
            "model_load_s": round(load320, 2), "steps": 10, "blocks": 3}
        m320.check_status()
        del m320, wav320, r320
        line["other_configs"] = other

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.arch, sd, clips_np, clip_s)
            line["speedup_vs_cpu"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
