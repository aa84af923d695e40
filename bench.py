#!/usr/bin/env python3
"""bench.py — audio-seconds/sec of WavTokenizer encode_infer + decode on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--arch hop600|hop320] [--clips 64]

One "step" = one encode_infer + decode round trip of `--clips` independent 3 s / 24 kHz clips
per GPU (BASELINE.json configs[1]: WavTokenizer-small-600, batch 64 x 3 s), inputs already
resident in HBM, synthetic clips and random-init weights of the real architecture
(wavtokenizer_amd/synth.py).  For N > 1 the driver launches one rank per GPU with
torch.distributed.run; clips shard across ranks with no collective on the data path (weak
scaling) and the per-step codes all-gather + waveform gather to rank 0 over RCCL is inside the
timed region.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     — dominant kernel (ConvNeXt pwconv1 GEMM + GELU, 12 launches per step): achieved =
                 algorithmic 2*M*N*K per launch / mean launch duration, from HIP events recorded on the
                 launch stream during the timed steps (wt_plan_set_timing).  The kernel evaluates every
                 fp32-equivalent product with THREE v_mfma_f32_32x32x16_f16 (split-f16, gemm16s.hip), so
                 its MFMA roofline in algorithmic (fp32-equivalent) FLOP/s is the dense f16 peak / 3.
  cpu_baseline — the oracle (oracle/cpu_ref.py, same ATen op sequence as the reference) timed on
                 this host's cores, rank 0 at N=1 only, on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2516.6     # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz)
PEAK_F16X3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0   # fp32-equivalent products on the f16 pipe: 3 MFMAs each
HBM_PEAK_GBS = 8000.0
CLIP_SECONDS = 3
SAMPLE_RATE = 24000
# algorithmic GFLOP per 3 s clip (2*MAC: convs, linears, LSTM, VQ, attention)
GFLOP_PER_CLIP = {"hop600": 20.852, "hop320": 38.910}
WORKLOAD = {"hop600": "WavTokenizer-small-600-24k-4096 (40 tok/s), encode_infer+decode round trip, 24 kHz clips",
            "hop320": "WavTokenizer-small-320-24k-4096 (75 tok/s), encode_infer+decode round trip, 24 kHz clips"}


def cpu_baseline(arch_name, sd, clips_np):
    """Oracle on the host cores.  torch's default thread count (= all cores) oversubscribes this
    small model, so a few thread counts are tried on a B=16 batch and the fastest is kept; B=1
    (BASELINE configs[0]) is timed at that count too.  ~30 s of CPU work in total."""
    from oracle.cpu_ref import OracleWavTokenizer
    from wavtokenizer_amd import NAMED_ARCHS
    orc = OracleWavTokenizer(NAMED_ARCHS[arch_name], sd)
    bw = torch.tensor([0])
    ncpu = os.cpu_count() or 1
    deadline = time.time() + 60.0

    def rate(B, reps):
        x = torch.from_numpy(clips_np[:B])
        with torch.inference_mode():
            f, _ = orc.encode_infer(x, bw)      # warm-up
            orc.decode(f, bw)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                f, _ = orc.encode_infer(x, bw)
                orc.decode(f, bw)
                ts.append(time.perf_counter() - t0)
                if time.time() > deadline:
                    break
        ts.sort()
        return B * CLIP_SECONDS / ts[len(ts) // 2]

    tried = {}
    for nt in sorted({min(ncpu, n) for n in (8, 16, 32, 64)}):     # all-cores (>64) only thrashes
        torch.set_num_threads(nt)
        tried[nt] = rate(16, 2)
    best_nt = max(tried, key=tried.get)
    torch.set_num_threads(best_nt)
    r16 = rate(16, 3)
    r1 = rate(1, 8)
    value = max(r16, r1)
    return {"value": round(value, 2), "unit": "audio-s/s", "cores": best_nt, "kind": "port",
            "sample": "oracle/cpu_ref.py (the reference's ATen op sequence, fp32); thread sweep on B=16 x3s: %s; at %d threads "
                      "median round trip B=16 x3s (3 reps): %.1f audio-s/s, B=1 x3s (8 reps): %.1f audio-s/s; value = the faster; "
                      "host has %d logical CPUs"
                      % (", ".join(f"{k}t={v:.1f}" for k, v in sorted(tried.items())), best_nt, r16, r1, ncpu)}


def pmc_traffic(arch_name, B):
    """Bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (separate
    FETCH_SIZE / WRITE_SIZE passes, gfx950-corrected: tools/summarize_profile.py); None if there is
    no summary for this workload.  PMC counters cannot be read from inside the timed process."""
    import glob
    if arch_name != "hop600" or B != 64 or CLIP_SECONDS != 3:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        kernels = json.load(f)["kernels"]
    for name, v in kernels.items():
        if "gemm16s_kernel" in name and name.rstrip().endswith(", 2, 1>(wt::GemmArgs)"):     # EPI_BIAS_GELU, S32 out
            return v["traffic_bytes_per_launch"]
    return None


def main():
    global CLIP_SECONDS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="hop600", choices=["hop600", "hop320"])
    ap.add_argument("--clips", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--clip-seconds", type=int, default=3, help="clip length (BASELINE configs[4] uses 30 s clips)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, synth, _capi
    arch = NAMED_ARCHS[args.arch]
    sd = synth.make_state_dict(arch, seed=0)
    model = WavTokenizer.from_arch(arch)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.eval().to(dev)
    CLIP_SECONDS = args.clip_seconds
    B, T = args.clips, CLIP_SECONDS * SAMPLE_RATE
    clips_np = synth.make_clips(B, T, seed=1000 * 2 + rank)       # seed = 1000*config + index
    wav = torch.from_numpy(clips_np).to(dev)
    bw = torch.tensor([0])
    L = arch.frames(T)

    from wavtokenizer_amd.sharding import gather_async

    # The end-of-step exchange (codes to every rank, 8*L bytes per clip; waveforms to rank 0, 18.4 MB per rank) is
    # issued asynchronously: RCCL moves step i's outputs over xGMI while step i+1's kernels run, and step i's result is
    # collected right before step i+1's exchange is issued (the last one before the closing barrier), so every
    # exchange is finished inside the timed region.
    pending = []

    def step():
        feats, codes = model.encode_infer(wav, bandwidth_id=bw)
        out = model.decode(feats, bandwidth_id=bw)
        if world > 1 and not args.no_gather:
            if args.backend == "gloo":                               # rehearsal only: through host memory
                codes, out = codes.cpu(), out.cpu()
            while pending:
                pending.pop().result()
            pending.append(gather_async(codes, out, dist, world, rank, dst=0))
        return codes, out

    def drain():
        res = None
        while pending:
            res = pending.pop().result()
        return res

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    # time the dominant kernel with HIP events on its launch stream during the timed steps
    dplan = model._engine.plans[(_capi.WT_PLAN_DECODE, B, L, model._graph_flags(B))][0]
    _capi.check(_capi.lib.wt_plan_set_timing(dplan, b"cnx.pwconv1"), "wt_plan_set_timing")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    tot_ms, n_l = ctypes.c_double(), ctypes.c_int64()
    _capi.check(_capi.lib.wt_plan_read_timing(dplan, ctypes.byref(tot_ms), ctypes.byref(n_l), 1), "wt_plan_read_timing")
    _capi.lib.wt_plan_set_timing(dplan, b"")

    # BASELINE configs[4] also asks for the p50 latency of one encode_infer call: a few synchronised calls after the
    # timed region (not part of `value`)
    p50_encode_ms = None
    if CLIP_SECONDS >= 30:
        lat = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model.encode_infer(wav, bandwidth_id=bw)
            e1.record()
            e1.synchronize()
            lat.append(e0.elapsed_time(e1))
        p50_encode_ms = sorted(lat)[len(lat) // 2]

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * CLIP_SECONDS * args.steps / elapsed
        Mrows = B * L
        flops = 2.0 * Mrows * arch.intermediate_dim * arch.dim
        kern_ms = tot_ms.value / max(1, n_l.value)
        achieved = flops / (kern_ms * 1e-3) / 1e12
        # SURVEY 8(d) FLOPs are for 3 s clips; attention grows with L^2 (4*L^2*768 per clip)
        gflop_clip = GFLOP_PER_CLIP[args.arch] * CLIP_SECONDS / 3.0 + 4.0 * arch.dim * (L * L - (L * 3 // CLIP_SECONDS) ** 2 * CLIP_SECONDS / 3.0) / 1e9
        e2e_tflops = gflop_clip * B * 1e9 / (ms_per_step * 1e-3) / 1e12
        line = {
            "metric": "audio-seconds/sec encode+decode, 24 kHz %d s clips" % CLIP_SECONDS,
            "value": round(value, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_note": "fp32 storage and accumulation; dense layers form each fp32 product from split f16 operands (3 MFMAs, error below the fp32 chain's own rounding)",
            "data": "synthetic",
            "config": {"workload": WORKLOAD[args.arch], "arch": args.arch, "clips_per_gpu": B,
                       "global_clips": world * B, "clip_seconds": CLIP_SECONDS, "frames_per_clip": L,
                       "codes_per_sec": round(world * B * L * args.steps / elapsed, 1),
                       **({"p50_encode_infer_ms_rank0": round(p50_encode_ms, 3)} if p50_encode_ms is not None else {}),
                       "weights": "random-init (synth seed 0)", "parallelism": f"clips sharded dp{world}",
                       "gather": ("codes all_gather + waveform gather to rank 0 (%s), asynchronous: overlaps the next step, all finished inside the timed region" % ("RCCL" if args.backend == "nccl" else "gloo rehearsal")) if world > 1 and not args.no_gather else "none"},
            "roofline": {"bound": "mfma",
                         "kernel": "wt::gemm16s_kernel<128,192,4,2,3,EPI_BIAS_GELU=2,OUT_S32=1> (ConvNeXt pwconv1 GEMM %dx%dx%d + GELU, "
                                   "split-f16: 3 x v_mfma_f32_32x32x16_f16 per fp32-equivalent product)" % (Mrows, arch.intermediate_dim, arch.dim),
                         "achieved": round(achieved, 2), "peak": round(PEAK_F16X3_TFLOPS, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F16X3_TFLOPS, 4), "traffic": pmc_traffic(args.arch, B),
                         "peak_note": "algorithmic fp32-equivalent FLOP/s; peak = %.1f TF dense f16 MFMA / 3 MFMAs per product; the kernel "
                                      "issues %.0f TF/s of f16 MFMA work" % (PEAK_F16_MFMA_TFLOPS, 3 * achieved),
                         "algorithmic_bytes_per_launch": 4 * (Mrows * arch.dim + arch.intermediate_dim * arch.dim + Mrows * arch.intermediate_dim),
                         "flops_per_launch": flops, "avg_launch_ms": round(kern_ms, 4), "launches_timed": n_l.value,
                         "end_to_end_tflops": round(e2e_tflops, 2),
                         "end_to_end_frac_of_fp32_mfma_peak": round(e2e_tflops / PEAK_FP32_MFMA_TFLOPS, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.arch, sd, clips_np)
            line["speedup_vs_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
