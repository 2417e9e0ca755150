"""End-to-end CLI on the GPU (tiny model, synthetic captions): experiment folder, results.csv, epoch_latest.pt in the
reference's checkpoint layout, resume, HF key round trip."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_train_cli_checkpoint_and_resume(tmp_path, monkeypatch):
    import torch
    import train_AT_text_only as cli
    monkeypatch.chdir(tmp_path)
    common = ["--model", "tiny-test-quickgelu", "--dataset-type", "synthetic", "--train-num-samples", "24", "--batch-size", "8",
              "--lr", "1e-4", "--wd", "1e-4", "--warmup", "2", "--rho", "6", "--k_adv", "1", "--seed", "1",
              "--custom_out_folder", "t_", "--logs", str(tmp_path / "logs"), "--name", "run"]
    assert cli.main(common + ["--epochs", "1"]) == 0
    out = tmp_path / "results" / "t_text_only_k1_rho6_seed1"
    ck = torch.load(out / "epoch_latest.pt", map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "name", "state_dict", "optimizer"} and ck["epoch"] == 1      # train_AT_text_only.py:516-525
    assert "transformer.resblocks.1.attn.in_proj_weight" in ck["state_dict"] and "logit_scale" in ck["state_dict"]
    # "optimizer" is torch.optim.AdamW.state_dict() in the reference's grouping (tests/test_checkpoint_cpu.py pins the layout)
    osd = ck["optimizer"]
    assert set(osd) == {"state", "param_groups"} and len(osd["param_groups"]) == 2 and 0 not in osd["state"]
    assert float(osd["state"][1]["step"]) == 3 and osd["param_groups"][0]["weight_decay"] == 0.0
    from leaf_amd.checkpoint import reference_param_groups
    g0, g1 = reference_param_groups(2)
    shapes = {k: tuple(v.shape) for k, v in ck["state_dict"].items()}
    params = {k: torch.nn.Parameter(torch.zeros(shapes[k])) for k in g0 + g1}
    ref_opt = torch.optim.AdamW([{"params": [params[k] for k in g0], "weight_decay": 0.0},
                                 {"params": [params[k] for k in g1], "weight_decay": 1e-4}], lr=1e-4)
    ref_opt.load_state_dict(osd)                       # what the reference's --resume does (train_AT_text_only.py:366)
    assert float(ref_opt.state[params["text_projection"]]["exp_avg"].abs().sum()) > 0
    rows = open(out / "results.csv").read().strip().splitlines()
    assert len(rows) == 2 and "loss" in rows[0]
    assert os.path.exists(tmp_path / "logs" / "run" / "params.txt")
    assert cli.main(common + ["--epochs", "1"]) == -1                                         # experiment exists
    assert cli.main(common[:-1] + ["run2", "--epochs", "2", "--resume", str(out / "epoch_latest.pt")]) == 0
    ck2 = torch.load(out / "epoch_latest.pt", map_location="cpu", weights_only=False)
    assert ck2["epoch"] == 2 and float(ck2["optimizer"]["state"][1]["step"]) == 6
    assert not torch.equal(ck["state_dict"]["text_projection"], ck2["state_dict"]["text_projection"])
    rows = open(out / "results.csv").read().strip().splitlines()
    assert len(rows) == 3 and rows[1].split(",")[0] in ("1", "1.0") and rows[2].split(",")[0] in ("2", "2.0"), "results.csv continues on resume"
    # --resume latest finds the checkpoint in the results folder (the reference looks where it never writes)
    assert cli.main(common[:-1] + ["run2", "--epochs", "3", "--resume", "latest"]) == 0
    assert torch.load(out / "epoch_latest.pt", map_location="cpu", weights_only=False)["epoch"] == 3


def test_full_clip_checkpoint_is_carried_through(tmp_path, monkeypatch):
    """A --pretrained FULL CLIP state_dict (visual.* included) comes back out of epoch_latest.pt complete, so the reference's
    strict model.load_state_dict (train_AT_text_only.py:363) and its converters accept it; --export-hf writes the release format."""
    import torch
    import train_AT_text_only as cli
    from leaf_amd.model import create_model
    monkeypatch.chdir(tmp_path)
    m = create_model("tiny-test-quickgelu", seed=3)
    full = {k: v.cpu() for k, v in m.state_dict().items()}
    full.update({"visual.conv1.weight": torch.randn(8, 3, 4, 4), "visual.proj": torch.randn(8, 64), "visual.ln_post.bias": torch.zeros(8)})
    torch.save(full, tmp_path / "clip_full.pt")
    args = ["--model", "tiny-test-quickgelu", "--pretrained", str(tmp_path / "clip_full.pt"), "--dataset-type", "synthetic",
            "--train-num-samples", "8", "--batch-size", "8", "--lr", "1e-4", "--wd", "1e-4", "--warmup", "2", "--rho", "4", "--k_adv", "1",
            "--seed", "4", "--epochs", "1", "--custom_out_folder", "f_", "--logs", str(tmp_path / "logs"), "--name", "runf",
            "--export-hf", str(tmp_path / "hf_out")]
    assert cli.main(args) == 0
    ck = torch.load(tmp_path / "results" / "f_text_only_k1_rho4_seed4" / "epoch_latest.pt", map_location="cpu", weights_only=False)
    assert set(full) == set(ck["state_dict"])
    assert torch.equal(ck["state_dict"]["visual.proj"], full["visual.proj"])
    assert not torch.equal(ck["state_dict"]["text_projection"], full["text_projection"])
    assert os.path.exists(tmp_path / "hf_out" / "model.safetensors") and os.path.exists(tmp_path / "hf_out" / "config.json")
    # the optimizer state has the reference's groups over the WHOLE CLIP (it builds AdamW before it freezes model.visual,
    # train_AT_text_only.py:326-341 vs :489-490): an AdamW built that way over this state_dict's parameters loads it
    from leaf_amd.checkpoint import reference_param_groups
    extra = {k: v for k, v in full.items() if k.startswith("visual.")}
    g0, g1 = reference_param_groups(2, extra)
    assert "visual.ln_post.bias" in g0 and "visual.conv1.weight" in g1 and "visual.proj" in g1
    osd = ck["optimizer"]
    assert [len(g["params"]) for g in osd["param_groups"]] == [len(g0), len(g1)]
    names = g0 + g1
    assert not any(names[i].startswith("visual.") or names[i] == "logit_scale" for i in osd["state"])
    params = {k: torch.nn.Parameter(torch.zeros(ck["state_dict"][k].shape)) for k in names}
    ref_opt = torch.optim.AdamW([{"params": [params[k] for k in g0], "weight_decay": 0.0},
                                 {"params": [params[k] for k in g1], "weight_decay": 1e-4}], lr=1e-4)
    ref_opt.load_state_dict(osd)
    assert float(ref_opt.state[params["token_embedding.weight"]]["step"]) == 1
    # ... and our own --resume reads it back (the visual.* names come from the same checkpoint's state_dict)
    args2 = [a for a in args if a not in ("--export-hf", str(tmp_path / "hf_out"))]
    args2[args2.index("runf")] = "runf2"
    args2[args2.index("--epochs") + 1] = "2"
    assert cli.main(args2 + ["--resume", "latest"]) == 0
    ck2 = torch.load(tmp_path / "results" / "f_text_only_k1_rho4_seed4" / "epoch_latest.pt", map_location="cpu", weights_only=False)
    assert ck2["epoch"] == 2 and [len(g["params"]) for g in ck2["optimizer"]["param_groups"]] == [len(g0), len(g1)]
    assert float(next(iter(ck2["optimizer"]["state"].values()))["step"]) == 2
    # --lock-image freezes the tower BEFORE the reference builds its optimizer (:286-290): text-only groups
    args3 = list(args2)
    args3[args3.index("runf2")] = "runf3"
    args3[args3.index("f_")] = "l_"
    args3[args3.index("--epochs") + 1] = "1"
    assert cli.main(args3 + ["--lock-image"]) == 0
    ck3 = torch.load(tmp_path / "results" / "l_text_only_k1_rho4_seed4" / "epoch_latest.pt", map_location="cpu", weights_only=False)
    t0, t1 = reference_param_groups(2)
    assert [len(g["params"]) for g in ck3["optimizer"]["param_groups"]] == [len(t0), len(t1)] and set(full) == set(ck3["state_dict"])


def test_hf_key_roundtrip_gives_same_embeddings():
    import torch
    from leaf_amd.checkpoint import openclip_to_hf
    from leaf_amd.model import LeafCLIPText, create_model, get_config
    from oracle import text_oracle as O
    m = create_model("tiny-test", seed=11)
    hf = openclip_to_hf(m.state_dict(), m.cfg)
    assert "text_model.encoder.layers.0.self_attn.q_proj.weight" in hf and hf["text_projection.weight"].shape == (64, 128)
    m2 = LeafCLIPText(get_config("tiny-test")).load_state_dict(hf)
    toks = O.synthetic_tokens(5, seed=1)
    assert np.array_equal(m.encode_text(toks).cpu().numpy(), m2.encode_text(toks).cpu().numpy())


def test_eval_textfare_script(tmp_path, monkeypatch):
    """eval_textfare.py equivalent (reference eval_textfare.py:112-149): clean loss is exactly 0 when the two models
    coincide, the attack can only raise the loss, and the CSV has the reference's columns."""
    import eval_textfare
    monkeypatch.chdir(tmp_path)
    (tmp_path / "caps.txt").write_text("a photo of a cat\nthe red car on the street\ntwo people in the park\nx\n")
    rows = eval_textfare.main(["--model", "tiny-test-quickgelu", "--texts", str(tmp_path / "caps.txt"), "--rho", "12",
                               "--k", "2", "--n-test", "4", "--batch-size", "3"])
    assert len(rows) == 4 and all(r["textfare_clean"] == 0.0 for r in rows)
    assert all(r["textfare_adv"] >= 0.0 for r in rows) and any(r["textfare_adv"] > 0.0 for r in rows)
    out = list((tmp_path / "results_textfare").glob("*_leaf_k2_rho_12.csv"))
    assert len(out) == 1 and out[0].read_text().splitlines()[0] == "sentence,adv_sentence,textfare_clean,textfare_adv"


def test_eval_textfare_per_sentence_replays_the_reference_loop(tmp_path, monkeypatch):
    """``--per-sentence`` = the reference's loop order (eval_textfare.py:113-141, one attack_text_leaf call per
    sentence): with the same numpy seed the RNG stream, the candidate strings and the adversarial sentences are the reference's own
    (fixture: tests/golden/make_golden_eval.py ran the reference's loop body on two tiny models).  Decisions are fp-level arg-maxes;
    where one flips, the engine's pick must be as good as the oracle's best within 16-bit noise (the margin rule of
    test_attack_text_replays_reference_trace)."""
    import json
    import eval_textfare
    from leaf_amd.tokenizer import SimpleTokenizer
    from oracle import text_oracle as O
    monkeypatch.chdir(tmp_path)
    with open(os.path.join(os.path.dirname(__file__), "golden", "eval_per_sentence.json")) as f:
        fx = json.load(f)
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    w = O.init_weights(cfg, seed=fx["model_seed"])
    tok = SimpleTokenizer()
    matched = 0
    for key, case in fx["cases"].items():
        sents = [r["sentence"] for r in case["rows"]]
        (tmp_path / "caps.txt").write_text("\n".join(sents) + "\n")
        trace = []
        rows = eval_textfare.main(["--model", fx["model"], "--texts", str(tmp_path / "caps.txt"), "--rho", str(case["rho"]), "--k", str(case["k"]),
                                   "--n-test", str(len(sents)), "--seed", str(case["seed"]), "--per-sentence",
                                   "--clean-seed", str(fx["clean_seed"]), "--robust-seed", str(fx["model_seed"])], trace=trace)
        assert len(rows) == len(trace) == len(sents)
        in_sync = True          # the global RNG is still where the reference's was at this sentence
        for got, stages, ref in zip(rows, trace, case["rows"]):
            assert abs(got["textfare_clean"] - ref["textfare_clean"]) < 5e-3 * ref["textfare_clean"]
            if not in_sync:
                continue
            assert stages[0] == ref["stage_candidates"][0], "stage-1 candidates differ: RNG stream / loop order drift"
            if got["adv_sentence"] == ref["adv_sentence"]:
                assert stages == ref["stage_candidates"]
                assert abs(got["textfare_adv"] - ref["textfare_adv"]) < 5e-3 * ref["textfare_adv"]
                matched += 1
                continue
            # a near-tie flipped.  If it was the LAST stage's arg-max (all stages scored the reference's candidates), the engine's pick
            # must be within 16-bit noise of the best candidate under the fp32 oracle; an earlier flip only counts against ``matched``
            if stages == ref["stage_candidates"]:
                anchor = O.encode_text(w, cfg, tok.encode_batch([got["sentence"]]))
                loss_o = ((O.encode_text(w, cfg, tok.encode_batch(stages[-1])) - anchor) ** 2).sum(-1)
                mine = ((O.encode_text(w, cfg, tok.encode_batch([got["adv_sentence"]])) - anchor) ** 2).sum(-1)[0]
                assert mine >= loss_o.max() * (1 - 5e-3), (got["adv_sentence"], ref["adv_sentence"], mine, loss_o.max())
            # with k > 1 a flipped decision changes the sentence length and with it the RNG consumption of what follows
            in_sync = case["k"] == 1
    assert matched >= 6, f"only {matched} of 8 sentences reproduce the reference's adversarial sentence"


def test_train_cli_normalize_fare_and_grad_clip(tmp_path, monkeypatch):
    """The two optional flags of the reference trainer that change the training arithmetic (--normalize_fare,
    utils_AT.py:296,319; --grad-clip-norm, utils_AT.py:348-357) run end to end and bound the loss / keep it finite."""
    import train_AT_text_only as cli
    monkeypatch.chdir(tmp_path)
    args = ["--model", "tiny-test-quickgelu", "--dataset-type", "synthetic", "--train-num-samples", "16", "--batch-size", "8",
            "--lr", "1e-4", "--wd", "1e-4", "--warmup", "2", "--rho", "6", "--k_adv", "1", "--seed", "2", "--epochs", "1",
            "--custom_out_folder", "n_", "--logs", str(tmp_path / "logs"), "--name", "runn",
            "--normalize_fare", "--grad-clip-norm", "1.0"]
    assert cli.main(args) == 0
    out = tmp_path / "results" / "n_text_only_k1_rho6_seed2"
    rows = open(out / "results.csv").read().strip().splitlines()
    hdr, vals = rows[0].split(","), rows[1].split(",")
    loss = float(vals[hdr.index("loss")])
    assert np.isfinite(loss) and 0.0 <= loss <= 4.0        # squared distance of two unit vectors
    # together with --accum-freq 2 the reference clips the running sum after EVERY micro-batch (utils_AT.py:348-362): under its
    # default --precision amp torch's GradScaler refuses the second unscale_ of a step -- same error here; without a scaler it runs
    acc = [a if a != "runn" else "runa" for a in args] + ["--accum-freq", "2"]
    acc[acc.index("--custom_out_folder") + 1] = "a_"
    with pytest.raises(RuntimeError, match="unscale_"):
        cli.main(acc)
    acc[acc.index("--custom_out_folder") + 1], acc[acc.index("--name") + 1] = "b_", "runb"
    assert cli.main(acc + ["--precision", "amp_bf16"]) == 0
    rows = open(tmp_path / "results" / "b_text_only_k1_rho6_seed2" / "results.csv").read().strip().splitlines()
    assert np.isfinite(float(rows[1].split(",")[rows[0].split(",").index("loss")]))


def test_train_cli_two_ranks_under_torch_distributed_run(tmp_path):
    """(The "--" in front of the script: torch.distributed.run's argparse rejects a script flag that is an ambiguous prefix of its
    own options -- the reference's --logs is one of --logs-specs / --logs_specs -- unless everything behind it is positional.)
    The trainer itself at world size 2: `python -m torch.distributed.run --nproc-per-node 2 train_AT_text_only.py
    --dist-backend gloo` with both ranks on this box's one GPU (RCCL needs a device per rank: the driver's scaling run), two epochs,
    --constrain and --accum-freq 2.  One checkpoint, the replicas' weights identical after every epoch (the CLI all-gathers a
    checksum), different captions and different search randomness per rank (seed + rank: train_AT_text_only.py:60-63,281), and a
    second launch into the same experiment folder ends on EVERY rank instead of hanging in the first collective."""
    import socket
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    words = tmp_path / "words.txt"
    words.write_text("\n".join("a photo of the small red car on wet street with two people near old house in sunny park at night "
                               "dog cat bird tree river bridge mountain snow beach city table chair".split()) + "\n")
    dump = tmp_path / "adv"
    dump.mkdir()

    def launch():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env["LEAF_DEBUG_DUMP_ADV"] = str(dump)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "--", os.path.join(root, "train_AT_text_only.py"), "--dist-backend", "gloo",
               "--model", "tiny-test-quickgelu", "--dataset-type", "synthetic", "--train-num-samples", "64", "--batch-size", "8",
               "--accum-freq", "2", "--lr", "1e-4", "--wd", "1e-4", "--warmup", "2", "--rho", "6", "--k_adv", "1", "--seed", "5",
               "--epochs", "2", "--constrain", "--dictionary-file", str(words), "--dictionary-tokenizer", "treebank",
               "--custom_out_folder", "d_", "--logs", str(tmp_path / "logs"), "--name", "run2", "--log-every-n-steps", "1"]
        return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=tmp_path)

    p = launch()
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    out = tmp_path / "results" / "d_text_only_k1_rho6_seed5"
    assert sorted(f.name for f in out.iterdir()) == ["epoch_latest.pt", "results.csv"]
    ck = torch.load(out / "epoch_latest.pt", map_location="cpu", weights_only=False)
    assert ck["epoch"] == 2 and float(ck["optimizer"]["state"][1]["step"]) == 4         # 64 / (8 * 2 ranks) / accum 2 = 2 steps per epoch
    log = p.stdout + p.stderr
    assert log.count("weights identical on 2 ranks") == 2, log[-3000:]
    a0, a1 = ((dump / f"adv_epoch0_rank{r}.txt").read_text().splitlines() for r in (0, 1))
    assert len(a0) == len(a1) == 8
    assert [l.split("\t")[0] for l in a0] != [l.split("\t")[0] for l in a1], "ranks must read different captions"
    assert any(l.split("\t")[0] != l.split("\t")[1] for l in a0 + a1), "the search changed nothing on either rank"
    # the same command again: the experiment exists -> the master says so and BOTH ranks leave (no hang, non-zero exit)
    p2 = launch()
    assert p2.returncode != 0 and "Experiment already exists" in p2.stdout + p2.stderr


def _write_datacomp_shard(path, first, n, img_bytes=3000):
    """A DataComp-style shard: per sample an image-sized member, the caption and a json record (data_AT.py:455-503 reads all
    three and drops the image; here only the .txt member is read)."""
    import io
    import tarfile
    with tarfile.open(path, "w", format=tarfile.USTAR_FORMAT) as tf:
        for i in range(first, first + n):
            for ext, payload in ((".jpg", os.urandom(img_bytes)),
                                 (".txt", f"a photo of the {['red', 'old', 'small', 'wet'][i % 4]} {['car', 'dog', 'house', 'tree', 'bridge'][i % 5]} number {i}\n".encode()),
                                 (".json", b'{"uid": "%d"}' % i)):
                ti = tarfile.TarInfo(f"{i:09d}{ext}")
                ti.size = len(payload)
                tf.addfile(ti, io.BytesIO(payload))


@pytest.mark.parametrize("model_name", ["tiny-test-quickgelu"])
def test_train_cli_from_tar_shards_with_and_without_the_reader_thread(tmp_path, monkeypatch, caplog, model_name):
    """SURVEY 8f-4 under the GPU suite (VERDICT r4 missing-1): the trainer fed by ``--dataset-type webdataset`` tar shards, as the
    reference's loop is by get_wds_dataset(..., tokenizer=None) (data_AT.py:455-503, consumed at utils_AT.py:282-290).  The
    captions reach the step in the order an independent ``tarfile`` walk of the (seed + epoch)-shuffled shard list gives; the
    background reader (--workers 1) and the in-loop reader (--workers 0) train on the same batches to the same loss and the same
    weights, bit for bit; a shard cut in the middle of a caption contributes what precedes the cut and is then skipped with the
    reference's log-and-continue warning (data_AT.py:285-288); the log line carries ``Load (t)``."""
    import logging
    import random
    import tarfile
    import torch
    import train_AT_text_only as cli
    import leaf_amd.train as T
    monkeypatch.chdir(tmp_path)
    shard_dir = tmp_path / "shards"
    shard_dir.mkdir()
    _write_datacomp_shard(shard_dir / "00000000.tar", 0, 11)
    _write_datacomp_shard(shard_dir / "00000001.tar", 100, 9)
    _write_datacomp_shard(shard_dir / "00000002.tar", 200, 6)
    whole = (shard_dir / "00000002.tar").read_bytes()
    # sample = 512 + 3072 (jpg) + 512 + 512 (txt) + 512 + 512 (json) = 5632 bytes; cut inside the SECOND sample's caption payload
    (shard_dir / "00000002.tar").write_bytes(whole[:5632 + 512 + 3072 + 512 + 20])
    pattern = str(shard_dir / "{00000000..00000003}.tar")             # ...3 does not exist: a missing shard is skipped too
    seed, bs, n = 7, 8, 32

    def expected_stream():
        shards = [str(shard_dir / f"{i:08d}.tar") for i in range(4)]
        random.Random(seed + 0).shuffle(shards)                        # TextLoader._stream: epoch 0
        while True:
            for p in shards:
                try:
                    with tarfile.open(p) as tf:
                        for m in tf:
                            if m.isfile() and m.name.endswith(".txt"):
                                data = tf.extractfile(m).read()
                                if len(data) < m.size:
                                    raise tarfile.ReadError("short member")
                                yield data.decode().strip()
                except (tarfile.TarError, OSError):
                    continue
    it = expected_stream()
    want = [[next(it) for _ in range(bs)] for _ in range(n // bs)]

    seen = {}
    real_attack = T.attack_text

    def run(workers, tag):
        seen[tag] = []

        def spy(model, tokenizer, texts, *a, **kw):
            seen[tag].append(list(texts))
            return real_attack(model, tokenizer, texts, *a, **kw)
        monkeypatch.setattr(T, "attack_text", spy)
        args = ["--model", model_name, "--train-data", pattern, "--dataset-type", "webdataset", "--train-num-samples", str(n),
                "--batch-size", str(bs), "--workers", str(workers), "--lr", "1e-4", "--wd", "1e-4", "--warmup", "2", "--rho", "6",
                "--k_adv", "1", "--seed", str(seed), "--epochs", "1", "--log-every-n-steps", "1",
                "--custom_out_folder", f"w{workers}_", "--logs", str(tmp_path / "logs"), "--name", f"run_w{workers}"]
        assert cli.main(args) == 0
        out = tmp_path / "results" / f"w{workers}_text_only_k1_rho6_seed{seed}"
        rows = open(out / "results.csv").read().strip().splitlines()
        loss = rows[1].split(",")[rows[0].split(",").index("loss")]
        ck = torch.load(out / "epoch_latest.pt", map_location="cpu", weights_only=False)
        return loss, ck["state_dict"]

    with caplog.at_level(logging.INFO):
        loss1, sd1 = run(1, "thread")
        loss0, sd0 = run(0, "inline")
    assert seen["thread"] == want, "captions out of order / not what tarfile reads"
    assert seen["inline"] == want
    assert any(c.endswith("number 200") for b in want for c in b) and not any(c.endswith("number 201") for b in want for c in b)
    assert loss1 == loss0 and float(loss1) > 0.0
    # same batches, same seed -> the same training run, up to the one place where the backward sums with atomics (tok_bwd scatters the
    # rows of a token that occurs several times in a batch with atomicAdd: the order of those fp32 additions is not fixed, and after
    # the first AdamW step the last bits of every later gradient follow)
    diff = {k: float((sd1[k].double() - sd0[k].double()).abs().max()) for k in sd1 if not torch.equal(sd1[k], sd0[k])}
    assert all(v <= 1e-6 for v in diff.values()), f"the reader thread changed the training result: {diff}"
    text = caplog.text
    assert text.count("Load (t):") >= 2 * (n // bs)
    assert "skipping shard" in text and "00000002.tar" in text and "00000003.tar" in text


def test_train_cli_two_ranks_with_the_sparse_embedding_reduction(tmp_path):
    """LEAF_DP_SPARSE_EMBED=1 through the trainer (one micro-batch per step, two ranks on this GPU, gloo): the replicas' weights are
    identical after the epoch (the CLI all-gathers a checksum) -- the token-embedding gradient travelled by touched rows."""
    import socket
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LEAF_DP_SPARSE_EMBED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--", os.path.join(root, "train_AT_text_only.py"), "--dist-backend", "gloo",
           "--model", "tiny-test-quickgelu", "--dataset-type", "synthetic", "--train-num-samples", "64", "--batch-size", "8",
           "--lr", "1e-4", "--wd", "1e-4", "--warmup", "2", "--rho", "6", "--k_adv", "1", "--seed", "5", "--epochs", "1",
           "--custom_out_folder", "s_", "--logs", str(tmp_path / "logs"), "--name", "runs", "--log-every-n-steps", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=tmp_path)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    log = p.stdout + p.stderr
    assert log.count("weights identical on 2 ranks") == 1, log[-3000:]
    ck = torch.load(tmp_path / "results" / "s_text_only_k1_rho6_seed5" / "epoch_latest.pt", map_location="cpu", weights_only=False)
    assert ck["epoch"] == 1 and float(ck["optimizer"]["state"][1]["step"]) == 4         # 64 / (8 * 2 ranks) = 4 steps
    emb = ck["state_dict"]["token_embedding.weight"]
    assert bool(torch.isfinite(emb).all())
