#!/usr/bin/env python3
"""Host-side ceiling of the folder driver for R ranks on ONE host (BASELINE.json configs[3]: 1 000 frames over 8 GPUs).

No 8-GPU node is available to this builder, so the GPU stage of predict.predict_folder is REPLACED BY A SLEEP of the
measured per-batch time and everything the host does per image runs for real, in R processes of T pool threads each
(the driver's shape: one process per GPU, NBC_HOST_WORKERS threads, predict.py:464-492):
    prepare:  read the .bmp, decode (24-bit BMP fast path), preprocess (a 1024 x 1024 frame passes through),
              write processed/<wood>/<name>.png
    (sleep):  one batch of `batch` frames every `gpu_ms_per_batch`
    finish:   label_png + write results/outputs/<wood>/<name>.png
A harness, not a product path: it imports the driver's own helpers and never touches a GPU.

  python scripts/host_ceiling.py [ranks=8] [threads=16] [images_per_rank=125] [gpu_images_per_s=270] [batch=2]

Prints per rank and in total: images/s, core-seconds per image and per stage, and which stage limits.  On a GPU box
whose CPU share is C cores the R x T threads share those C cores: the figure to carry over to an 8-GPU host is the
core-seconds per image (host work per image), from which images/s = cores / core-seconds per image."""
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def rank_main(rank, threads, n, gpu_ips, batch, src_dir, out_root, q):
    from concurrent.futures import ThreadPoolExecutor
    from neuralbarkcalculator_amd import predict as drv
    from neuralbarkcalculator_amd.pngio import write_png
    names = sorted(os.listdir(src_dir))
    woods = drv.WOOD_TYPES
    for kind in ("processed", "outputs"):
        for w in woods:
            os.makedirs(os.path.join(out_root, "r%d" % rank, kind, w), exist_ok=True)
    lvl_p, lvl_l = drv._png_level("processed"), drv._png_level("labels")
    stage = {"decode": 0.0, "preprocess": 0.0, "write_processed": 0.0, "write_labels": 0.0}
    cpu0 = time.process_time()

    def prepare(i):
        t0 = time.perf_counter()
        img = drv._decode_rgb(os.path.join(src_dir, names[i % len(names)]))
        t1 = time.perf_counter()
        out = drv.preprocess_image(img, 1024)
        t2 = time.perf_counter()
        write_png(os.path.join(out_root, "r%d" % rank, "processed", woods[i % 3], "f%04d.png" % i), out, lvl_p)
        t3 = time.perf_counter()
        stage["decode"] += t1 - t0; stage["preprocess"] += t2 - t1; stage["write_processed"] += t3 - t2
        return out

    def finish(i, lab):
        t0 = time.perf_counter()
        write_png(os.path.join(out_root, "r%d" % rank, "outputs", woods[i % 3], "f%04d.png" % i), drv.label_png(lab), lvl_l)
        stage["write_labels"] += time.perf_counter() - t0

    # label maps with the statistics of network output: large regions of the three classes
    yy, xx = np.mgrid[0:1024, 0:1024]
    labs = [((np.sin(xx / (40.0 + 7 * k)) + np.cos(yy / (55.0 + 5 * k))) > 0.6).astype(np.uint8) +
            ((np.sin(xx / 23.0 + k) * np.cos(yy / 31.0)) > 0.8).astype(np.uint8) for k in range(4)]
    gpu_s_per_batch = batch / gpu_ips
    t_start = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        window = 64
        futs = {i: pool.submit(prepare, i) for i in range(min(window, n))}
        done = []
        gpu_free_at = time.perf_counter()
        waited = 0.0
        for a in range(0, n, window):
            for i in range(a + window, min(a + 2 * window, n)):
                futs[i] = pool.submit(prepare, i)
            for b in range(a, min(a + window, n), batch):
                part = list(range(b, min(b + batch, a + window, n)))
                t0 = time.perf_counter()
                for i in part:
                    futs.pop(i).result()
                waited += time.perf_counter() - t0
                # the "GPU": batches run back to back, one every gpu_s_per_batch
                now = time.perf_counter()
                gpu_free_at = max(gpu_free_at, now) + gpu_s_per_batch
                if gpu_free_at - now > 4 * gpu_s_per_batch:          # four batches in flight at most, like the driver
                    time.sleep(gpu_free_at - now - 4 * gpu_s_per_batch)
                for i in part:
                    done.append(pool.submit(finish, i, labs[i % 4]))
        for f in done:
            f.result()
        time.sleep(max(0.0, gpu_free_at - time.perf_counter()))
    wall = time.perf_counter() - t_start
    q.put({"rank": rank, "wall_s": wall, "images": n, "cpu_s": time.process_time() - cpu0, "stage_s": stage,
           "main_waited_for_frames_s": waited})


def main():
    a = sys.argv[1:]
    ranks = int(a[0]) if len(a) > 0 else 8
    threads = int(a[1]) if len(a) > 1 else 16
    n = int(a[2]) if len(a) > 2 else 125
    gpu_ips = float(a[3]) if len(a) > 3 else 270.0
    batch = int(a[4]) if len(a) > 4 else 2
    from PIL import Image
    from neuralbarkcalculator_amd import synth
    root = tempfile.mkdtemp(prefix="nbc_host_")
    try:
        src = os.path.join(root, "src")
        os.makedirs(src)
        for i in range(40):                       # 40 distinct 1024 x 1024 frames, shared by the ranks (page cache warm, as in time_folder.py's second run)
            Image.fromarray(synth.make_frame(i, 1024, 1024), mode="RGB").save(os.path.join(src, "f%04d.bmp" % i))
        cores = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                cores = min(cores, int(quota) / int(period))
        except (OSError, ValueError):
            pass
        q = mp.Queue()
        t0 = time.perf_counter()
        procs = [mp.Process(target=rank_main, args=(r, threads, n, gpu_ips, batch, src, os.path.join(root, "out"), q)) for r in range(ranks)]
        for p in procs:
            p.start()
        res = sorted((q.get() for _ in procs), key=lambda d: d["rank"])
        for p in procs:
            p.join()
        wall = time.perf_counter() - t0
        total = sum(d["images"] for d in res)
        cpu = sum(d["cpu_s"] for d in res)
        print("host ceiling: %d ranks x %d threads, %d images each, GPU stage = sleep at %.0f images/s per rank (batch %d); cores available %.1f"
              % (ranks, threads, n, gpu_ips, batch, cores))
        for d in res:
            print("  rank %d: %.1f images/s, %.3f core-s per image; thread-seconds per image: %s; main thread waited %.2f s for frames"
                  % (d["rank"], d["images"] / d["wall_s"], d["cpu_s"] / d["images"],
                     ", ".join("%s %.4f" % (k, v / d["images"]) for k, v in d["stage_s"].items()), d["main_waited_for_frames_s"]))
        slowest = max(d["wall_s"] for d in res)
        print("  all ranks: %d images in %.2f s (slowest rank %.2f s) = %.1f images/s; %.3f core-s per image -> %.0f images/s per 16 cores, "
              "%.0f on 8 x 16 cores; GPU side asks %.0f images/s" % (total, wall, slowest, total / slowest, cpu / total, 16.0 / (cpu / total),
                                                                    128.0 / (cpu / total), ranks * gpu_ips))
        print("  limiting stage here: %s" % ("the host cores (%.1f of them busy on average)" % (cpu / slowest) if total / slowest < 0.9 * ranks * gpu_ips
                                             else "the (simulated) GPUs"))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
