#!/usr/bin/env python3
"""Where does one rank's image preparation stop scaling with workers?  (the 8-rank soak sits at ~400
prepared images/s per rank whatever the number of PIL threads.)  One process, Food-101 image sizes, noise images:

  stage x workers table for THREADS (full prepare_image / JPEG round trip / bicubic resize / to-array), then the full stage on worker
  PROCESSES, then the thread table again with glibc told to keep big blocks on its heap (mallopt: no mmap / munmap per image - every
  munmap of a multi-threaded process interrupts all its running threads for the TLB flush and takes the address-space lock).

  python tools/probe_host_prep_scaling.py [--n 512] [--workers 1,4,8,16,32]
"""
from __future__ import annotations

import argparse
import ctypes
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
from pathlib import Path

import numpy as np
from PIL import Image

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import DATASET_SIZES  # noqa: E402
from lmms_owc_amd.models import imageproc  # noqa: E402

MAXP, MINP = 1024 * 28 * 28, 4 * 28 * 28
_r = np.random.default_rng(0)
SIZES = DATASET_SIZES["food101"](_r, 64)
BASE = [Image.fromarray(_r.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB") for h, w in SIZES]


def full(i):
    return imageproc.prepare_image(BASE[i % 64], MINP, MAXP).shape


def jpeg(i):
    return imageproc.jpeg_round_trip(BASE[i % 64]).size


def resize(i):
    return BASE[i % 64].resize((448, 448), Image.BICUBIC).size


def to_array(i):
    return np.ascontiguousarray(np.asarray(BASE[i % 64], dtype=np.uint8).transpose(2, 0, 1)).shape


def table(tag: str, workers: list[int], n: int, stages) -> None:
    for name, fn in stages:
        out = []
        for t in workers:
            with ThreadPoolExecutor(t) as ex:
                list(ex.map(fn, range(4 * t)))
                t0 = time.perf_counter()
                list(ex.map(fn, range(n)))
                out.append(f"{t:3d} thr {n / (time.perf_counter() - t0):7.1f}/s")
        print(f"{tag:18s} {name:9s}", "  ".join(out), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--workers", default="1,4,8,16,32")
    args = ap.parse_args()
    workers = [int(x) for x in args.workers.split(",")]
    stages = [("full", full), ("jpeg", jpeg), ("resize", resize), ("to_array", to_array)]
    print(f"# {os.cpu_count()} cpus, affinity {len(os.sched_getaffinity(0))}, Pillow {Image.__version__}, "
          f"OWC_JPEG_BYTESIO={os.environ.get('OWC_JPEG_BYTESIO', '0')}", flush=True)
    table("threads", workers, args.n, stages)
    for p in [w for w in workers if w > 1]:
        with ProcessPoolExecutor(p) as ex:
            list(ex.map(full, range(4 * p)))
            t0 = time.perf_counter()
            list(ex.map(full, range(args.n), chunksize=4))
            print(f"{'processes':18s} {'full':9s} {p:3d} proc {args.n / (time.perf_counter() - t0):7.1f}/s", flush=True)
    from lmms_owc_amd.models._base import keep_image_blocks_mapped

    for p in [w for w in workers if w > 1]:
        with ProcessPoolExecutor(p, initializer=keep_image_blocks_mapped) as ex:
            list(ex.map(full, range(8 * p)))
            t0 = time.perf_counter()
            list(ex.map(full, range(args.n), chunksize=4))
            print(f"{'processes+keep':18s} {'full':9s} {p:3d} proc {args.n / (time.perf_counter() - t0):7.1f}/s", flush=True)
    libc = ctypes.CDLL("libc.so.6")
    M_TRIM_THRESHOLD, M_MMAP_THRESHOLD, M_ARENA_MAX = -1, -3, -8
    assert libc.mallopt(M_MMAP_THRESHOLD, 1 << 30) == 1 and libc.mallopt(M_TRIM_THRESHOLD, 1 << 30) == 1
    table("threads+mallopt", workers, args.n, stages[:2])
    Image.core.set_blocks_max(512)
    table("..+pillow blocks", workers, args.n, stages[:2])


if __name__ == "__main__":
    main()
