#!/usr/bin/env python3
"""Captions per second out of webdataset-style shards with image-sized dummy members.

Writes N_SHARDS tar files of SAMPLES samples each ({key}.jpg of ~IMG_KB random bytes, {key}.txt, {key}.json, the layout img2dataset
gives DataComp shards) under a scratch directory, then times (1) the header-scanning reader, (2) Python's tarfile on the same
shards, (3) a TextLoader epoch with the background reader while the consumer sleeps STEP_MS per batch (a stand-in for the training
step): the time the consumer waits per batch is what the trainer logs as ``Load (t)``.

    python tools/loader_bench.py [--shards 4 --samples 2000 --img-kb 60 --batch 128 --step-ms 25]
"""
import argparse
import io
import os
import sys
import tarfile
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_shards(root, shards, samples, img_kb, seed=0):
    import random
    rng = random.Random(seed)
    words = "a photo of the small red car on wet street with two people near old house".split()
    paths = []
    blob = os.urandom(img_kb * 1024)
    for s in range(shards):
        p = os.path.join(root, f"{s:08d}.tar")
        with tarfile.open(p, "w", format=tarfile.USTAR_FORMAT) as tf:
            for i in range(samples):
                key = f"{s:05d}{i:05d}"
                for ext, data in ((".jpg", blob[: img_kb * 1024 - rng.randint(0, 4096)]),
                                  (".txt", " ".join(rng.choice(words) for _ in range(rng.randint(3, 14))).encode()),
                                  (".json", b'{"width": 512, "height": 512}')):
                    ti = tarfile.TarInfo(key + ext)
                    ti.size = len(data)
                    tf.addfile(ti, io.BytesIO(data))
        paths.append(p)
    return paths


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=4)
    ap.add_argument("--samples", type=int, default=2000)
    ap.add_argument("--img-kb", type=int, default=60)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--step-ms", type=float, default=25.0)
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    from leaf_amd import train
    with tempfile.TemporaryDirectory(dir=a.dir) as root:
        t0 = time.time()
        paths = write_shards(root, a.shards, a.samples, a.img_kb)
        total = a.shards * a.samples
        size = sum(os.path.getsize(p) for p in paths)
        print(f"{a.shards} shards x {a.samples} samples, {size / 1e6:.0f} MB written in {time.time() - t0:.1f} s")
        t0 = time.time()
        n = sum(1 for p in paths for _ in train._scan_tar_captions(p))
        dt = time.time() - t0
        print(f"header scan      : {n} captions in {dt:.3f} s = {n / dt:,.0f} captions/s")
        t0 = time.time()
        n2 = 0
        for p in paths:
            with tarfile.open(p) as tf:
                for m in tf:
                    if m.isfile() and m.name.endswith(".txt"):
                        tf.extractfile(m).read()
                        n2 += 1
        dt = time.time() - t0
        print(f"tarfile (before) : {n2} captions in {dt:.3f} s = {n2 / dt:,.0f} captions/s")
        assert n == n2 == total
        for prefetch in (0, 4):
            ld = train.TextLoader(None, paths, a.batch, total, 0, 1, seed=1, prefetch=prefetch)
            waited, t_all, nb = 0.0, time.time(), 0
            it = iter(ld)
            while True:
                t0 = time.time()
                try:
                    next(it)
                except StopIteration:
                    break
                waited += time.time() - t0
                nb += 1
                time.sleep(a.step_ms / 1e3)
            print(f"TextLoader prefetch={prefetch}: {nb} batches of {a.batch}, consumer waited {1e3 * waited / nb:.3f} ms per batch "
                  f"beside a {a.step_ms:g}-ms step ({time.time() - t_all:.2f} s in all)")


if __name__ == "__main__":
    main()
