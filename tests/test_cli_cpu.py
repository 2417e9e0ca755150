"""Host logic of the CLI drop-in on CPU: flag set, Adam defaults, schedule, shard expansion, caption readers."""
import io
import json
import os
import tarfile

import numpy as np
import pytest

from leaf_amd import params as P
from leaf_amd import train as T

REF_SCRIPT_FLAGS = """--save-frequency 1 --zeroshot-frequency 1 --report-to wandb --wandb-project-name datacomp_small
 --train-data path/{00000000..00000003}.tar --imagenet-val=path/to/imagenet/val --val-text-classification fancyzhx/ag_news
 --warmup 1400 --batch-size=128 --accum-freq=1 --lr=1e-5 --wd=1e-4 --epochs=30 --workers=8 --model hf-hub:chs20/fare2-clip
 --dataset-type webdataset --train-num-samples 80000 --val-num-samples 1024 --k_adv 1 --k_adv_test 1 --rho=50
 --n_charmer_test=20 --n_val_imagenet 1000 --seed 1 --custom_out_folder ViT-L-FARE2_constrained_ --constrain""".split()


def test_reference_launch_script_flags_parse():
    a = P.parse_args(REF_SCRIPT_FLAGS)     # scripts/train_leaf_vitl.sh
    assert (a.batch_size, a.rho, a.k_adv, a.lr, a.wd, a.warmup, a.epochs, a.seed) == (128, 50, 1, 1e-5, 1e-4, 1400, 30, 1)
    assert a.constrain and a.custom_out_folder == "ViT-L-FARE2_constrained_"
    # params_AT.py:17-23,600-604: "vit" is not in "hf-hub:chs20/fare2-clip" -> beta2 0.999 / eps 1e-8
    assert (a.beta1, a.beta2, a.eps) == (0.9, 0.999, 1e-8)
    b = P.parse_args(["--model", "hf-hub:laion/CLIP-ViT-H-14-laion2B-s32B-b79K"])
    assert (b.beta2, b.eps, b.lr, b.rho, b.wd, b.warmup) == (0.98, 1e-6, 5e-4, 20, 0.2, 10000)


def test_cosine_lr_matches_reference_values(golden_dir):
    c = json.load(open(os.path.join(golden_dir, "manifest.json")))["cosine_lr"]

    class Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0}]
    sched = T.cosine_lr(Opt, c["base_lr"], c["warmup"], c["steps"])
    for s, v in c["values"].items():
        sched(int(s))
        assert abs(Opt.param_groups[0]["lr"] - v) < 1e-12 and Opt.param_groups[1]["lr"] == Opt.param_groups[0]["lr"]


def test_brace_expansion_and_tar_caption_reader(tmp_path):
    assert T._expand_braces("a/{0008..0011}.tar") == [f"a/{i:04d}.tar" for i in range(8, 12)]
    assert T._expand_braces("x/{1..2}.tar::y/{7..7}.tar") == ["x/1.tar", "x/2.tar", "y/7.tar"]
    shard = tmp_path / "00000000.tar"
    with tarfile.open(shard, "w") as tf:
        for i in range(6):
            for ext, payload in ((".jpg", b"\xff\xd8not-an-image"), (".txt", f"caption number {i}".encode())):
                ti = tarfile.TarInfo(f"s{i:03d}{ext}")
                ti.size = len(payload)
                tf.addfile(ti, io.BytesIO(payload))
    caps = list(T._iter_tar_captions([str(shard), str(tmp_path / "missing.tar")]))   # missing shard is skipped
    assert caps == [f"caption number {i}" for i in range(6)]
    args = P.parse_args(["--train-data", str(tmp_path / "{00000000..00000000}.tar"), "--dataset-type", "webdataset",
                         "--train-num-samples", "6", "--batch-size", "3"])
    args.rank, args.world_size = 0, 1
    data = T.get_text_data(args)
    batches = list(data["train"].dataloader)
    assert len(batches) == 2 and all(b[0] is None and len(b[1]) == 3 for b in batches)


def test_synthetic_and_text_datasets(tmp_path):
    f = tmp_path / "caps.txt"
    f.write_text("a cat\n\na dog\nthe red car\n")
    args = P.parse_args(["--train-data", str(f), "--batch-size", "2"])
    args.rank, args.world_size = 0, 1
    d = T.get_text_data(args)["train"].dataloader
    assert d.num_samples == 3 and d.num_batches == 2
    args = P.parse_args(["--dataset-type", "synthetic", "--train-num-samples", "10", "--batch-size", "4"])
    args.rank, args.world_size = 0, 1
    d = T.get_text_data(args)["train"].dataloader
    assert d.num_batches == 3 and all(isinstance(t, str) and t for _, b in d for t in b)


def test_lr_schedules_match_the_reference_known_answers(golden_dir):
    """--lr-scheduler cosine | const | const-cooldown (train_AT_text_only.py:384-401) against values produced by the reference's
    own open_clip_train/scheduler.py (tests/golden/make_golden_sched.py); both param groups are assigned; an unknown name and a
    cooldown without --epochs-cooldown are refused as the reference refuses them."""
    import json
    import os
    import types
    import pytest
    from leaf_amd import train as T
    with open(os.path.join(golden_dir, "sched_kat.json")) as f:
        k = json.load(f)

    class Opt:
        def __init__(self):
            self.param_groups = [{"lr": 0.0}, {"lr": 0.0}]

    def args(**kw):
        base = dict(lr=k["base_lr"], warmup=k["warmup"], accum_freq=1, epochs_cooldown=None, lr_cooldown_power=1.0, lr_cooldown_end=0.0)
        base.update(kw)
        return types.SimpleNamespace(**base)
    nb = 50                                          # batches per epoch: 4 epochs = 200 steps, 1 cooldown epoch = 50 steps
    makers = {"cosine": args(lr_scheduler="cosine"), "const": args(lr_scheduler="const"),
              "const-cooldown p=1 end=0": args(lr_scheduler="const-cooldown", epochs_cooldown=1),
              "const-cooldown p=2 end=1e-6": args(lr_scheduler="const-cooldown", epochs_cooldown=1, lr_cooldown_power=2.0, lr_cooldown_end=1e-6)}
    for c in k["cases"]:
        o = Opt()
        f = T.make_scheduler(makers[c["name"]], o, k["total_steps"], nb)
        for s_, want in zip(k["steps"], c["values"]):
            got = f(s_)
            assert abs(got - want) <= 1e-18 + 1e-12 * abs(want), (c["name"], s_, got, want)
            assert o.param_groups[0]["lr"] == got and o.param_groups[1]["lr"] == got
    with pytest.raises(ValueError, match="Unknown scheduler"):
        T.make_scheduler(args(lr_scheduler="linear"), Opt(), 200, nb)
    with pytest.raises(AssertionError, match="cooldown epochs"):
        T.make_scheduler(args(lr_scheduler="const-cooldown"), Opt(), 200, nb)


def _write_shard(path, n, fmt=tarfile.USTAR_FORMAT, mode="w", long_names=False, img_bytes=3000):
    with tarfile.open(path, mode, format=fmt) as tf:
        for i in range(n):
            key = ("very_long_directory_name_" * 6 + f"/s{i:03d}") if long_names else f"s{i:03d}"
            for ext, payload in ((".jpg", os.urandom(img_bytes)), (".txt", f"  caption {i} of {os.path.basename(str(path))} \n".encode()),
                                 (".json", b"{}")):
                ti = tarfile.TarInfo(key + ext)
                ti.size = len(payload)
                tf.addfile(ti, io.BytesIO(payload))


def test_header_scan_reader_equals_tarfile_and_falls_back(tmp_path):
    """The .txt-only header scan returns what tarfile returns; archives it does not understand (gzip, pax /
    GNU long names) take the tarfile path; truncated and missing shards are skipped with a warning (data_AT.py:285-288)."""
    plain, gz, pax, gnu = (tmp_path / n for n in ("a.tar", "b.tar.gz", "c.tar", "d.tar"))
    _write_shard(plain, 7)
    _write_shard(gz, 3, mode="w:gz")
    _write_shard(pax, 3, fmt=tarfile.PAX_FORMAT, long_names=True)
    _write_shard(gnu, 3, fmt=tarfile.GNU_FORMAT, long_names=True)
    assert list(T._scan_tar_captions(str(plain))) == [f"caption {i} of a.tar" for i in range(7)]
    for p in (gz, pax, gnu):
        with pytest.raises(T._NotPlainTar):
            list(T._scan_tar_captions(str(p)))
    want = [f"caption {i} of {p.name}" for p, n in ((plain, 7), (gz, 3), (pax, 3), (gnu, 3)) for i in range(n)]
    assert list(T._iter_tar_captions([str(plain), str(gz), str(pax), str(gnu)])) == want
    cut = tmp_path / "cut.tar"
    cut.write_bytes(plain.read_bytes()[:5 * 512 + 100])      # ends inside the second sample
    got = list(T._iter_tar_captions([str(cut), str(tmp_path / "missing.tar"), str(plain)]))
    assert got[-7:] == want[:7] and len(got) <= 9


def test_prefetching_loader_yields_the_synchronous_batches(tmp_path):
    """one background reader + bounded queue; same batches in the same order as the synchronous loader, an early
    break stops the reader thread, and a reader-side error surfaces in the consumer instead of hanging it."""
    import threading
    shards = []
    for s in range(3):
        p = tmp_path / f"{s:08d}.tar"
        _write_shard(p, 11)
        shards.append(str(p))
    sync = T.TextLoader(None, shards, 4, 40, 0, 1, seed=3, prefetch=0)
    pre = T.TextLoader(None, shards, 4, 40, 0, 1, seed=3, prefetch=2)
    a, b = list(sync), list(pre)
    assert a == b and len(a) == 10 and all(len(t) == 4 for _, t in a)
    pre.set_epoch(1); sync.set_epoch(1)
    assert list(pre) == list(sync) and list(pre) != a          # reshuffled per epoch (seed + epoch), still identical
    before = threading.active_count()
    it = iter(pre)
    next(it)
    it.close()                                                  # consumer leaves early: the finally block joins the reader
    assert threading.active_count() == before
    empty = T.TextLoader(None, [str(tmp_path / "nothing.tar")], 4, 8, 0, 1, seed=0, prefetch=2)
    with pytest.raises(RuntimeError, match="no caption"):
        list(empty)
    args = P.parse_args(["--train-data", shards[0], "--dataset-type", "webdataset", "--train-num-samples", "8", "--batch-size", "4", "--workers", "0"])
    args.rank, args.world_size = 0, 1
    assert T.get_text_data(args)["train"].dataloader.prefetch == 0
