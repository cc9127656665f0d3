#!/usr/bin/env python3
"""Throughput bench of the hot path: 1024x1024 images/s through libnbc_hip.so.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.  For N > 1 the driver may
launch it through ``python -m torch.distributed.run --nproc-per-node N ...`` (one rank per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment); started WITHOUT that environment,
``bench.py --gpus N`` starts the N ranks itself (a fresh ``torch.distributed.run`` child process,
spawned before this process touches the GPU) and relays rank 0's line.  A *step* is one pass of the
hot path (model call + argmax + class counts, models.py:269-270,324-331) over one batch of synthetic
frames that are already resident in HBM.  W untimed warm-up steps, then exactly K steps bracketed by
barrier + torch.cuda.synchronize() on both sides; the time is the MAX over ranks; rank 0 prints ONE
JSON line.

Headline (``value``, ``dtype`` "f16x2"): BASELINE.json configs[1] -- 1 x MI355X per rank, batch 1,
synthetic 1024x1024x3 frames, in the f32-GRADE mode NBC_PREC_F16X2: every f32 value (activation or weight) is kept as
two f16 pieces (to 2^-23 relative for |x| >= 2^-12; weight rows normalised by a power of two per output channel), a
product is three EXACT f16 products on v_mfma_f32_16x16x32_f16, sums are f32 in two levels.  It passes every test of the f32 mode under the SAME tolerances
(tests/test_gpu_parity.py, tests/test_gpu_configs.py: logits within 5e-6 of the oracle's logit range, at most 4 label
flips per megapixel, each adjudicated by float64); its error against a float64 evaluation is at the f32 MFMA mode's level
(4.2-5.2e-6 on logits of range 2-3.5, the f32 MFMA 4.0-6.2e-6, the CPU reference 3.6-4.5e-6:
profiles/r04_fp64_adjudication_*).  The f32 MFMA mode (v_mfma_f32_32x32x2_f32),
round 2's headline, rides along as ``f32_mfma_batch1``.
Images are independent: N GPUs = N shards of the folder, no data-path collective,
weak scaling; the collectives are the one-off RCCL broadcast of the packed weights (``setup``) and
the all_gather of the per-image rows at the end (``gather_s``), both outside the timed region.

Extra objects on the same line:
  ``roofline``      dominant kernel of the headline run (the implicit-GEMM convolution, MFMA-bound):
                    algorithmic FLOPs / RAW HIP-event durations of its launches, measured in a second
                    region of K steps on one stream with an event between launches (nothing subtracted)
  ``cpu_baseline``  the torch-CPU oracle on this box's host cores, a bounded sample (rank 0, N = 1)
  ``parity``        label match of the headline run against the oracle on sample frames
  ``f32_mfma_batch1`` configs[1] in f32 activations and weights on v_mfma_f32_32x32x2_f32: own timed region, value,
                    roofline and parity
  ``bf16_batch8``   BASELINE.json configs[2] (batch 8, bf16 throughput mode) as its own object: its own
                    timed region (same bracket), value, roofline and parity -- never the headline
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# dense MFMA peaks, MI355X_MICROARCH.md; f16x2 spends three f16 MFMAs per f32 product: its peak in f32-equivalent
# (algorithmic) FLOPs is a third of the f16 peak
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "f16x2": 2500.0 / 3.0}
HBM_MEASURED = 6.29e12                          # achievable HBM bytes/s, same guide
H = W = 1024
DTYPE_NAME = {"fp32": "f32", "bf16": "bf16", "f16x2": "f16x2"}
DTYPE_NOTE = {
    "fp32": "f32 activations and weights, v_mfma_f32_32x32x2_f32 (exact f32 fma chain), two-level f32 sums",
    "bf16": "bf16 activations and weights, f32 accumulate, f32 BN epilogue: throughput mode, not f32 grade",
    "f16x2": "f32-grade: each f32 value as two f16 pieces (to 2^-23 relative for |x| >= 2^-12, an absolute 2^-36 below; weight rows "
             "normalised by a power of two per output channel, folded into the f32 BN scale), each product = 3 exact f16 products on "
             "v_mfma_f32_16x16x32_f16 (the dropped 4th is 2^-22 relative at worst), two-level f32 sums; same test tolerances as f32, "
             "error against float64 at the f32 MFMA mode's level (profiles/r04_fp64_adjudication_*)"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--precision", choices=["bf16", "fp32", "f16x2"], default="f16x2",
                    help="arithmetic of the HEADLINE run: f16x2 (f32-grade on the f16 matrix pipe) or fp32 (f32 MFMA), the two "
                         "modes that pass the f32 tolerances; bf16 only for A/B experiments")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--streams", type=int, default=2,
                    help="independent forwards kept in flight on separate HIP streams (each its own workspace)")
    ap.add_argument("--weights", choices=["random_init", "trained_like"], default="trained_like")
    ap.add_argument("--frames", type=int, default=100,
                    help="distinct synthetic frames resident in HBM and cycled through by the steps (configs[1] names 100: "
                         "1.26 GB as f32 NCHW)")
    ap.add_argument("--parity-frames", type=int, default=4,
                    help="distinct frames the parity objects compare with the CPU oracle (the bf16 batch-8 leg uses 8)")
    ap.add_argument("--conv-tile", type=int, default=-1, help="-1 = per-layer choice, 0.. force a tile shape (A/B runs)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real multi-GPU path) or gloo (rehearsal)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (one-GPU box), implies a gloo process group")
    ap.add_argument("--autotune", choices=["auto", "on", "off"], default="auto",
                    help="per-layer tiles by measurement in the set-up (nbc_autotune) or the library's default choice (its cost "
                         "model): auto = default tiles for the f16x2 headline (318 images/s against 313 with measured tiles, two "
                         "forwards in flight) and measured tiles for the f32 MFMA and bf16 legs, as in rounds 2 and 3")
    ap.add_argument("--no-autotune", action="store_true", help="same as --autotune off")
    ap.add_argument("--tune-objective", choices=["auto", "latency", "throughput"], default="auto",
                    help="what nbc_autotune minimises: a launch's time alone, or time x share of the chip it occupies "
                         "(auto: throughput when several forwards are in flight)")
    ap.add_argument("--tune-reps", type=int, default=3, help="timed launches per tile and layer in nbc_autotune")
    ap.add_argument("--dump-ops", default=None, help="write the headline run's per-launch records (JSON) to this file")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the configs[2] object (batch 8, bf16)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the f32_mfma_batch1 object (configs[1] on the f32 MFMA)")
    ap.add_argument("--f32-steps", type=int, default=0, help="steps of the f32 MFMA leg (default: max(10, K/4))")
    ap.add_argument("--f32-streams", type=int, default=4)
    ap.add_argument("--bf16-steps", type=int, default=0, help="steps of the configs[2] leg (default: max(10, K/5))")
    ap.add_argument("--bf16-streams", type=int, default=2)
    ap.add_argument("--no-op-events", action="store_true", help="no instrumented region (no roofline object)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="how often the headline's bracketed K-step region is run: `value` is the FIRST (the contract's region), "
                         "`value_repeats` lists all of them, `value_median` their median (the line's own spread)")
    ap.add_argument("--save-tiles", default=None, help="write the measured per-layer tile choices (JSON) to this file")
    ap.add_argument("--tiles-file", default=None,
                    help="install the tile choices of an earlier run instead of measuring them (profiling runs: the trace "
                         "then holds no autotune launches)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """--gpus N without a torchrun environment: start the N ranks as a CHILD process (this process has
    not touched the GPU and never will), relay their output and return the child's exit code."""
    import torch
    if not args.share_gpu:
        have = torch.cuda.device_count()          # counts devices without initialising HIP
        if have < args.gpus:
            print("bench.py: --gpus %d but this node shows %d GPU(s); use --share-gpu for a one-GPU rehearsal"
                  % (args.gpus, have), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def host_cores():
    """CPU share of this process: cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("NBC_CPU_BASELINE_THREADS", "16")))   # GPU-box share per GPU is 16


def kernel_source_id():
    """sha256 over the kernel sources: a PMC traffic file is only quoted for the kernels it measured."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "neuralbarkcalculator_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    import torch
    from neuralbarkcalculator_amd import synth
    from neuralbarkcalculator_amd.model import FCNResNet50

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d: refusing to print a line for another job size"
                  % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    # stdout carries rank 0's ONE JSON line and nothing else: whatever the libraries print on it while they load,
    # connect or run (gloo announces its peers, RCCL its version) goes to stderr; the line is written to the saved
    # descriptor at the very end
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    dist = None
    setup = {}
    t0 = time.perf_counter()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_gpu:
            local_rank = 0
            args.dist_backend = "gloo"
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
        dist.barrier()
        setup["process_group_init_s"] = time.perf_counter() - t0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    sd = synth.make_state_dict(args.weights, seed=7) if rank == 0 else None
    nf = max(1, args.frames)
    # frames: each rank owns its shard of the folder (rank r takes global images r, r+world, ...)
    # (generated on a few host threads: 0.3-0.5 s per frame on one)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, host_cores())) as pool:
        frames = list(pool.map(lambda i: synth.make_input(rank + world * i, H, W), range(nf)))
    frames_dev = None                                # the same frames in HBM, shared by the legs

    def run_leg(precision, batch, nstreams, steps, warmup, instrument, repeats=1):
        """One configuration: weights to every rank, workspace, per-layer tiles, W warm-up steps, the
        bracketed region of K steps, then (optionally) the instrumented region."""
        leg = {"setup": {}}
        t0 = time.perf_counter()
        model = FCNResNet50(precision)
        if rank == 0:
            model.load_state_dict(sd)
        model.to(dev)
        torch.cuda.synchronize()
        leg["setup"]["pack_and_upload_s"] = time.perf_counter() - t0
        if world > 1:
            barrier()
            t0 = time.perf_counter()
            model.broadcast_weights(src=0)
            torch.cuda.synchronize()
            leg["setup"]["weight_broadcast_s"] = max_over_ranks(time.perf_counter() - t0)
            leg["setup"]["weight_blob_bytes"] = int(model._blob_dev.numel())
        t0 = time.perf_counter()
        nonlocal frames_dev
        if frames_dev is None:
            frames_dev = [torch.from_numpy(f).to(dev) for f in frames]
        if batch == 1:
            batches = [f[None] for f in frames_dev]                  # views: nothing is copied
        else:                                        # batch i = frames i, i+1, ... (cyclically), stacked on the device
            batches = [torch.stack([frames_dev[(i + j) % nf] for j in range(batch)]) for i in range(min(nf, steps + warmup))]
        model.reserve(batch, H, W)
        model.set_conv_tile(args.conv_tile)
        tiles = None
        key = "%s_b%d_s%d" % (precision, batch, nstreams)
        installed = None
        if args.tiles_file:
            installed = json.load(open(args.tiles_file))
        want_tune = {"on": True, "off": False, "auto": precision != "f16x2"}["off" if args.no_autotune else args.autotune]
        tune = want_tune and args.conv_tile < 0 and installed is None
        objective = ("throughput" if nstreams > 1 else "latency") if args.tune_objective == "auto" else args.tune_objective
        if tune:
            tiles = model.autotune(batches[0], reps=args.tune_reps, objective=objective)   # setup: per-layer tile shape by measurement
        elif installed is not None:
            tiles = installed[key]
            model.set_plan_tiles(tiles)
        else:
            tiles = model.plan_tiles()                # the library's default choice (or the forced tile where it fits)
        leg["tiles_source"] = "nbc_autotune (%s objective)" % objective if tune else "installed from a file" if installed is not None else "library default (cost model)"
        models = [model]
        for _ in range(nstreams - 1):
            m2 = model.clone_shared()
            m2.reserve(batch, H, W)
            m2.set_conv_tile(args.conv_tile)
            if tune:
                m2.autotune(batches[0], reps=args.tune_reps, objective=objective)
            elif installed is not None:
                m2.set_plan_tiles(tiles)
            models.append(m2)
        leg["tiles_key"] = key
        streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(nstreams - 1)]
        torch.cuda.synchronize()
        leg["setup"]["workspace_and_autotune_s"] = time.perf_counter() - t0

        def step(i, k=None):
            k = i % nstreams if k is None else k
            with torch.cuda.stream(streams[k]):
                return models[k].predict_labels(batches[i % len(batches)], labels_dtype=torch.uint8)

        def timed_region(n_steps, k=None):
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_steps):
                step(i, k)
            torch.cuda.synchronize()
            barrier()
            return max_over_ranks(time.perf_counter() - t0)

        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        dt = timed_region(steps)                      # (1) the contract's region: nothing but the hot path in it
        # the same region again (each with its own barrier + synchronize bracket): the spread of the line itself
        leg["dt_repeats"] = [dt] + [timed_region(steps) for _ in range(max(1, repeats) - 1)]
        leg.update(dt=dt, steps=steps, warmup=warmup, batch=batch, nstreams=nstreams, tiles=tiles,
                   model=model, precision=precision)
        # per-image rows (global_idx, H, W, count_1, count_2) of this rank's last step, as the folder driver gathers them
        labels, counts = step(steps - 1, 0)
        torch.cuda.synchronize()
        leg["last_counts"] = counts.cpu().numpy()
        # (2) instrumented region: the same K steps on ONE stream with a HIP event between launches
        # (recorded on the forward's stream) and tiles tuned for a launch running alone.  RAW durations.
        if instrument:
            if tune and nstreams > 1:
                leg["tiles_one_stream"] = model.autotune(batches[0], objective="latency")
            elif installed is not None and nstreams > 1 and key + "_one_stream" in installed:
                model.set_plan_tiles(installed[key + "_one_stream"])
            for i in range(min(warmup, 3)):
                step(i, 0)
            dt_single = timed_region(steps, 0)
            model.set_profiling(True)
            dt_events = timed_region(steps, 0)
            leg["records"] = model.op_records()
            model.set_profiling(False)
            leg["dt_single"] = dt_single
            leg["dt_events"] = dt_events
        for m in models[1:]:
            m._destroy()
        return leg

    def roofline_of(leg):
        prec = leg["precision"]
        records = leg["records"]
        steps = leg["steps"]
        conv = [r for r in records if r["kernel"] == "conv_dma"]
        dom = [r for r in conv if r["cout"] % 128 == 0 and r["name"] != "backbone.conv1"]   # the wide-tile instantiations
        flops = sum(r["flops"] for r in dom)
        ms = sum(r["ms"] for r in dom)
        c3 = [r for r in conv if r["k"] == 3]
        ach = flops / (ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[prec]
        bound_s = sum(max(r["flops"] / (peak * 1e12), r["bytes"] / HBM_MEASURED) for r in records)
        out = {
            "bound": "mfma",
            "kernel": "conv_dma_kernel<%s>%s (LDS-DMA implicit GEMM; every convolution with Cout %% 128 == 0, stem excluded)"
                      % (prec, " + conv3x3_rowstep_kernel (its 3x3 layers of 128-pixel-wide maps: rows resident in LDS, one barrier per row-step)" if prec == "f16x2" else ""),
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
            "measured_on": "second region of K steps, one stream, a HIP event between launches on the forward's stream, "
                           "latency-tuned tiles; RAW event durations (each includes its event packet, nothing subtracted)",
            "launches_per_step": len(dom), "flops_per_launch": flops / len(dom), "avg_launch_ms": ms / len(dom),
            "algorithmic_bytes_per_launch": sum(r["bytes"] for r in dom) / len(dom),
            "conv3x3_tflops": sum(r["flops"] for r in c3) / (sum(r["ms"] for r in c3) * 1e-3) / 1e12,
            "all_conv_tflops": sum(r["flops"] for r in conv) / (sum(r["ms"] for r in conv) * 1e-3) / 1e12,
            "sum_kernel_ms_per_step": sum(r["ms"] for r in records),
            "event_free_ms_per_step_one_stream": 1e3 * leg["dt_single"] / steps,
            "instrumented_ms_per_step": 1e3 * leg["dt_events"] / steps,
            "event_cost_us_per_launch_estimate": max(0.0, (leg["dt_events"] - leg["dt_single"]) / steps / len(records)) * 1e6,
            "layerwise_bound_ms_per_step": 1e3 * bound_s,
            "layerwise_frac_timed_region": bound_s / (leg["dt"] / steps),
        }
        out["conv3x3_frac"] = out["conv3x3_tflops"] / peak
        if prec == "f16x2":
            out["peak_note"] = ("FLOPs are the convolution's algorithmic (f32-equivalent) FLOPs; the kernel issues three f16 MFMA FLOPs "
                                "per algorithmic FLOP, so the peak is a third of the 2.5 PF dense f16 peak and frac = matrix-pipe "
                                "share of that peak")
            out["mfma_flops_per_algorithmic_flop"] = 3
            out["achieved_f16_mfma_tflops"] = 3.0 * ach
        # Fabric traffic of the dominant kernel comes from rocprofv3 PMC passes of this same configuration
        # (scripts/profile_round.sh -> scripts/per_forward_table.py); quoted only when the table was made for
        # this batch, precision and kernel source.
        path = os.path.join(ROOT, "profiles", "per_forward_ops_%s_b%d.json" % (DTYPE_NAME[prec], leg["batch"]))
        try:
            t = json.load(open(path))
            if t.get("batch") == leg["batch"] and t.get("precision") == prec and t.get("kernel_source_id") == kernel_source_id():
                out["traffic"] = t["dominant_kernel"]["traffic_bytes_per_launch"]
                # the same kernel's figure by rocprofv3 --kernel-trace (medians over the forwards of the same command): the
                # raw HIP events of `frac` include an event packet per launch (about 3 us on launches of 70-170 us)
                if "frac_of_peak" in t["dominant_kernel"]:
                    out["frac_rocprof_table"] = t["dominant_kernel"]["frac_of_peak"]
                    out["table"] = "profiles/%s" % os.path.basename(path)
                    # the longest launch of the table (the head conv, classifier.0): in-kernel shader clock and matrix-pipe
                    # utilisation of its PMC pass -- where the power limit shows (busy share x clock)
                    hc = max((o for o in t.get("ops", []) if o.get("flops", 0) > 0), key=lambda o: o["median_us"], default=None)
                    if hc is not None and "clock_ghz_in_pmc_pass" in hc:
                        out["head_conv_rocprof_table"] = {"op": hc["op"], "median_us": hc["median_us"], "clock_ghz_in_pmc_pass": hc["clock_ghz_in_pmc_pass"],
                                                          "mfma_util": hc.get("mfma_util"), "frac_of_peak": hc["tflops"] / peak}
                out["traffic_note"] = ("rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per launch of the dominant kernel (separate PMC passes, "
                                       "L2-miss traffic incl. Infinity-Cache hits), from profiles/%s" % os.path.basename(path))
                if "mfma_util" in t["dominant_kernel"]:     # matrix-pipe utilisation by hardware counters, same table
                    out["mfma_util_by_counters"] = t["dominant_kernel"]["mfma_util"]
                    out["mfma_util_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs) over the dominant kernel's launches "
                                             "(rocprofiler-sdk's MfmaUtil: busy cycles per shader cycle at the clock the launches ran at; GRBM_GUI_ACTIVE over-counts on launches "
                                             "under 0.3 ms, so the figure reads low against FLOPs over time there; the head conv alone: see the table)")
            else:
                out["traffic_note"] = "profiles/%s was measured for another batch / precision / kernel source: not quoted" % os.path.basename(path)
        except (OSError, KeyError, ValueError):
            out["traffic_note"] = "no PMC pass for this configuration in profiles/"
        worst = sorted(records, key=lambda r: -r["ms"])[:6]
        out["top_ops"] = [{"name": r["name"], "ms": round(r["ms"], 4),
                           "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1) if r["ms"] > 0 else 0} for r in worst]
        return out

    # ---- headline: configs[1] in the reference's arithmetic
    head = run_leg(args.precision, args.batch, max(1, args.streams), args.steps, args.warmup, not args.no_op_events, args.repeats)
    # ---- configs[1] on the f32 MFMA (round 2's headline), when the headline is the f16x2 mode
    legf = None
    if not args.no_f32_leg and (args.precision, args.batch) == ("f16x2", 1):
        sf = args.f32_steps if args.f32_steps > 0 else max(10, args.steps // 4)
        legf = run_leg("fp32", 1, max(1, args.f32_streams), sf, max(2, min(args.warmup, 4)), not args.no_op_events)
    # ---- configs[2]: batch 8, bf16
    leg8 = None
    if not args.no_bf16_leg:
        s8 = args.bf16_steps if args.bf16_steps > 0 else max(10, args.steps // 5)
        leg8 = run_leg("bf16", 8, max(1, args.bf16_streams), s8, max(2, min(args.warmup, 4)), not args.no_op_events)

    # ---- the gather of the per-image rows (one all_gather of int64 rows, predict.py's collective)
    gather_s = None
    if dist is not None:
        from neuralbarkcalculator_amd.predict import gather_rows
        rows = np.array([[rank + world * j, H, W, int(head["last_counts"][j, 1]), int(head["last_counts"][j, 2])]
                         for j in range(args.batch)], dtype=np.int64)
        barrier()
        t0 = time.perf_counter()
        allrows = gather_rows(rows, world * args.batch, world, dist, dev if args.dist_backend == "nccl" else None)
        gather_s = max_over_ranks(time.perf_counter() - t0)
        assert len(allrows) == world * args.batch

    if rank == 0 and args.save_tiles:
        saved = {}
        for leg in (head, legf, leg8):
            if leg is not None and leg["tiles"] is not None:
                saved[leg["tiles_key"]] = leg["tiles"]
                if "tiles_one_stream" in leg:
                    saved[leg["tiles_key"] + "_one_stream"] = leg["tiles_one_stream"]
        with open(args.save_tiles, "w") as f:
            json.dump(saved, f)
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    cfg = "configs[%d]" % (1 if args.precision in ("fp32", "f16x2") and args.batch == 1 else 2 if (args.precision, args.batch) == ("bf16", 8) else 0)
    if cfg == "configs[0]":
        cfg = "A/B run (not a BASELINE.json config)"
    images = world * args.batch * args.steps
    out = {
        "metric": "1024x1024 images/sec (whole node) + per-pixel label match vs CPU ref",
        "value": images / head["dt"],
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * head["dt"] / args.steps,
        # the bracketed K-step region run `--repeats` times back to back: `value` is the first, these are all of them
        "value_repeats": [images / d for d in head["dt_repeats"]],
        "value_median": images / sorted(head["dt_repeats"])[len(head["dt_repeats"]) // 2],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE_NAME[args.precision],
        "dtype_note": DTYPE_NOTE[args.precision],
        "data": "synthetic",
        "config": {"workload": "%s: 1xMI355X per rank, batch=%d, %s arithmetic, fcn_resnet50 (%s weights, eval), synthetic "
                               "1024x1024x3 frames resident in HBM, forward + argmax + class counts"
                               % (cfg, args.batch, DTYPE_NAME[args.precision] + (" (f32-grade)" if args.precision == "f16x2" else ""), args.weights),
                   "batch": args.batch, "frames": nf, "precision": args.precision, "shard": "images r, r+N, ... per rank",
                   "conv_tile": args.conv_tile, "streams": head["nstreams"],
                   "tiles": head["tiles"], "tiles_source": head["tiles_source"]},
        "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
        "dist_backend": (args.dist_backend if dist is not None else None),
        "devices": "cuda:0 shared by every rank (rehearsal)" if args.share_gpu else "cuda:LOCAL_RANK, one GPU per rank",
        "setup": dict(setup, **head["setup"]),
        "gather_s": gather_s,
    }
    if "records" in head:
        out["roofline"] = roofline_of(head)
        out["top_ops"] = out["roofline"].pop("top_ops")
        if args.dump_ops:
            with open(args.dump_ops, "w") as f:
                json.dump(head["records"], f, indent=1)

    ref_cache = {}

    def oracle_on(frame_idx):
        if frame_idx not in ref_cache:
            from oracle.fcn_resnet50_oracle import OracleFCNResNet50, predict_labels
            if "model" not in ref_cache:
                torch.set_num_threads(host_cores())
                m = OracleFCNResNet50()
                m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
                ref_cache["model"] = m
            ref_cache[frame_idx] = predict_labels(ref_cache["model"], torch.from_numpy(frames[frame_idx])[None])
        return ref_cache[frame_idx]

    def parity_of(model, frame_ids, batch):
        """Labels and low-res logits of `model` on the given distinct frames, run in batches of `batch` (the last one
        padded by cycling), against the CPU oracle frame by frame."""
        res = {"frames": "synthetic frames %s of rank 0, run in batches of %d" % (list(frame_ids), batch),
               "precision": model.precision, "pixels": 0, "label_mismatches": 0, "max_oracle_margin_at_mismatch": 0.0,
               "max_lowres_logit_err_over_oracle_range": 0.0, "frames_compared": 0, "per_frame_label_mismatches": []}
        for a in range(0, len(frame_ids), batch):
            chunk = list(frame_ids[a:a + batch])
            ids = [chunk[j % len(chunk)] for j in range(batch)]
            x = torch.stack([frames_dev[i] for i in ids])
            labels, counts, lowres = model.predict_labels(x, return_lowres=True)
            torch.cuda.synchronize()
            for j, i in enumerate(chunk):
                labels_ref, counts_ref, logits_ref, lowres_ref = oracle_on(i)
                res["max_lowres_logit_err_over_oracle_range"] = max(
                    res["max_lowres_logit_err_over_oracle_range"],
                    float((lowres[j].cpu() - lowres_ref[0]).abs().max()) / float(logits_ref.abs().max()))
                bad = labels[j].cpu() != labels_ref[0]
                top2 = torch.topk(logits_ref, 2, dim=1).values
                margin = (top2[:, 0] - top2[:, 1])[0]
                res["pixels"] += int(bad.numel())
                res["label_mismatches"] += int(bad.sum())
                res["per_frame_label_mismatches"].append(int(bad.sum()))
                res["frames_compared"] += 1
                if bool(bad.any()):
                    res["max_oracle_margin_at_mismatch"] = max(res["max_oracle_margin_at_mismatch"], float(margin[bad].max()))
                res["oracle_logit_range"] = float(logits_ref.abs().max())
                res.setdefault("oracle_class_counts", counts_ref[0].tolist())
                res.setdefault("gpu_class_counts", counts[j].cpu().tolist())
        res["label_match"] = 1.0 - res["label_mismatches"] / max(1, res["pixels"])
        return res

    n_par = max(1, min(args.parity_frames, nf))
    if world == 1 and not args.no_cpu_baseline:
        # the oracle (a port of the reference's torch-CPU forward, eval mode) + argmax on this box's host
        # cores: 1 warm-up + 3 timed calls on ONE 1024x1024 frame (about 10-30 s of CPU work)
        from oracle.fcn_resnet50_oracle import predict_labels
        oracle_on(0)
        x = torch.from_numpy(frames[0])[None]
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            predict_labels(ref_cache["model"], x)
            times.append(time.perf_counter() - t0)
        med = sorted(times)[1]
        out["cpu_baseline"] = {"value": 1.0 / med, "unit": "images/s", "cores": host_cores(), "kind": "port",
                               "sample": "1 synthetic 1024x1024 frame, torch %s CPU oracle (eval, f32) + argmax, 1 warm-up + "
                                         "median of 3" % torch.__version__,
                               "s_per_image": med}
    if world == 1 and not args.no_parity:
        out["parity"] = parity_of(head["model"], list(range(n_par)), args.batch)
    if legf is not None:
        of = {
            "config": {"workload": "configs[1]: 1xMI355X per rank, batch=1, f32 activations/weights on v_mfma_f32_32x32x2_f32 (round 2's "
                                   "headline), synthetic 1024x1024x3 frames resident in HBM, forward + argmax + class counts",
                       "batch": 1, "precision": "fp32", "streams": legf["nstreams"], "tiles": legf["tiles"], "tiles_source": legf["tiles_source"]},
            "value": world * legf["steps"] / legf["dt"], "unit": "images/s", "dtype": "f32", "dtype_note": DTYPE_NOTE["fp32"],
            "steps": legf["steps"], "warmup": legf["warmup"], "ms_per_step": 1e3 * legf["dt"] / legf["steps"],
            "setup": legf["setup"],
        }
        if "records" in legf:
            of["roofline"] = roofline_of(legf)
            of["top_ops"] = of["roofline"].pop("top_ops")
        if world == 1 and not args.no_parity:
            of["parity"] = parity_of(legf["model"], list(range(n_par)), 1)
        out["f32_mfma_batch1"] = of
    if leg8 is not None:
        o8 = {
            "config": {"workload": "configs[2]: 1xMI355X per rank, batch=8, bf16 activations/weights (f32 accumulate, fused f32 "
                                   "BN+ReLU epilogue), synthetic 1024x1024x3 frames resident in HBM, forward + argmax + class counts",
                       "batch": 8, "precision": "bf16", "streams": leg8["nstreams"], "tiles": leg8["tiles"], "tiles_source": leg8["tiles_source"]},
            "value": world * 8 * leg8["steps"] / leg8["dt"], "unit": "images/s", "dtype": "bf16",
            "steps": leg8["steps"], "warmup": leg8["warmup"], "ms_per_step": 1e3 * leg8["dt"] / leg8["steps"],
            "setup": leg8["setup"],
        }
        if "records" in leg8:
            o8["roofline"] = roofline_of(leg8)
            o8["top_ops"] = o8["roofline"].pop("top_ops")
        if world == 1 and not args.no_parity:
            o8["parity"] = parity_of(leg8["model"], list(range(min(nf, max(8, n_par)))), 8)
        out["bf16_batch8"] = o8
    sys.stdout.flush()
    os.write(line_fd, (json.dumps(out) + "\n").encode())
    os.close(line_fd)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
