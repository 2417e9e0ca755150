"""CPU-only: tokenizer / mutation / constraint known answers from the reference, and the C-ABI loads with
every symbol that include/leaf_hip.h declares."""
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tokenizer_known_answers(golden_dir):
    from leaf_amd.tokenizer import SimpleTokenizer
    with open(os.path.join(golden_dir, "tokenizer_kat.json")) as f:
        kat = json.load(f)
    tok = SimpleTokenizer()
    ids = tok.encode_batch(kat["texts"])
    assert ids.dtype == np.int32 and ids.shape == (len(kat["texts"]), 77)
    assert np.array_equal(ids, np.array(kat["ids"], dtype=np.int32))
    t = tok(kat["texts"][:2])
    assert str(t.dtype) == "torch.int64" and tuple(t.shape) == (2, 77)
    assert tok.encode_batch("a photo of a cat")[0, :7].tolist() == [49406, 320, 1125, 539, 320, 2368, 49407]
    assert tok.decode(tok.encode("a photo of a cat")).strip() == "a photo of a cat"


def test_generate_sentence_known_answers(golden_dir):
    from leaf_amd import attacks
    with open(os.path.join(golden_dir, "mutation_kat.json")) as f:
        kat = json.load(f)
    assert kat["V"] == attacks.DEFAULT_V
    for c in kat["generate_sentence"]:
        assert attacks.generate_sentence(c["S"], c["z"], c["u"], kat["V"], 1, alternative=c["alt"]) == c["out"], c
    for c in kat["space_all"]:
        assert attacks.generate_all_sentences(c["S"], [ord(' ')], subset_z=None, alternative=-1) == c["out"]
    np.random.seed(7)
    for c in kat["random_at_z_seed7"]:
        assert attacks.generate_random_sentences_at_z(c["S"], c["z"], kat["V"], c["n"], alternative=-1) == c["out"]


def test_constraint_rule(golden_dir):
    from leaf_amd import attacks
    with open(os.path.join(golden_dir, "mutation_kat.json")) as f:
        kat = json.load(f)
    attacks.set_dictionary(attacks.Dictionary(kat["stub_words"]))
    try:
        got = attacks.valid_sentence_batched(
            ["a photo of a cat", "the red car"],
            [["a photo of a ca t", "a photo of acat", "a photo of a cat", "a phot o of a cat"],
             ["thered car", "the red ca r", "the re d car", "the red car"]])
        assert got == kat["valid_batched"]
    finally:
        attacks.set_dictionary(None)


def test_c_abi_exports_every_declared_symbol():
    from leaf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "leaf_hip.h")).read()
    declared = set(re.findall(r"\b(leaf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"leaf_text_cfg"}
    l = _lib.lib()
    for name in declared:
        assert hasattr(l, name), f"{name} declared in include/leaf_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert l.leaf_version() >= 1
    # the product library carries no experiment and no diagnostic export: those live in the builds under
    # tools/diag/ (include/leaf_hip_diag.h, leaf_amd/csrc/variants/)
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (leaf_[A-Za-z0-9_]+)", syms))
    assert exported == declared, exported ^ declared
    assert "gemm128pp" not in syms and "leaf_debug" not in syms
    diag = open(os.path.join(ROOT, "include", "leaf_hip_diag.h")).read()
    assert set(re.findall(r"\b(leaf_debug_[a-z0-9_]+)\s*\(", diag)) == set(_lib.DIAG_SIGS)


def test_handle_layout_and_errors_without_gpu():
    """Host-only parts of the ABI: config validation, flat layout (decay group first), sizes."""
    import ctypes as C
    from leaf_amd import _lib
    l = _lib.lib()
    h = C.c_void_p()
    bad = _lib.TextCfgC(2, 100, 2, 64, 77, 49408, 0, 1e-5)
    assert l.leaf_text_create(C.byref(bad), 1, C.byref(h)) != 0 and b"unsupported" in l.leaf_last_error()
    cfg = _lib.TextCfgC(12, 768, 12, 768, 77, 49408, 1, 1e-5)
    assert l.leaf_text_create(C.byref(cfg), 1, C.byref(h)) == 0
    assert l.leaf_text_param_count(h) == 123650304           # ViT-L text tower parameters
    name = C.create_string_buffer(128)
    off, rows, cols = C.c_size_t(), C.c_int64(), C.c_int64()
    seen, total = [], 0
    for i in range(l.leaf_text_num_tensors(h)):
        assert l.leaf_text_param_info(h, i, name, 128, C.byref(off), C.byref(rows), C.byref(cols)) == 0
        assert off.value == total
        n = name.value.decode()
        numel = rows.value * (cols.value or 1)
        excluded = cols.value == 0 or "bn" in n or "ln" in n or "bias" in n or "logit_scale" in n
        assert excluded == (off.value >= l.leaf_text_decay_count(h)), n   # train_AT_text_only.py:323-331
        total += numel
        seen.append(n)
    assert total == 123650304 and len(set(seen)) == len(seen) == 12 * 12 + 5
    assert l.leaf_text_workspace_bytes(h, 6400, 1) > 6400 * 768 * 4
    assert l.leaf_text_param_info(h, 999, name, 128, None, None, None) != 0
    # split masks (round 6): argument checks happen before anything touches the device
    masks = (C.c_int32 * 4)(3, 2, 2, 2)
    fake = C.c_void_p(64)                      # never dereferenced on these paths
    assert l.leaf_text_split_bytes(h, 4) == ((4 * 27 * 768 * 768 * 2 + 255) // 256) * 256 + 4 * 7 * 768 * 4
    assert l.leaf_text_split_pack_masks(h, fake, masks, 12, fake, None) != 0 and b"out of range" in l.leaf_last_error()
    assert l.leaf_text_split_pack_masks(h, None, masks, 4, fake, None) != 0
    bad_masks = (C.c_int32 * 2)(3, 16)
    assert l.leaf_text_split_pack_masks(h, fake, bad_masks, 2, fake, None) != 0 and b"bits 0..3" in l.leaf_last_error()
    assert l.leaf_text_split_pack_masks(h, fake, masks, 0, None, None) == 0 and l.leaf_text_get_option(h, b"split_blocks") == 0
    assert l.leaf_text_set_option(h, b"ln_fold", 0) == 0
    assert l.leaf_text_split_pack_masks(h, fake, masks, 4, fake, None) != 0 and b"LN-folded" in l.leaf_last_error()
    assert l.leaf_text_get_option(h, b"compact_resid") == 1 and l.leaf_text_get_option(h, b"no_such_option") == -1
    assert l.leaf_text_precise_workspace_bytes(h, 128) > 128 * 77 * 768 * 4 * 9
    l.leaf_text_destroy(h)


def test_every_kernel_sets_the_fp16_saturation_mode_first():
    """F16::pack2 (common.h) leaves the +-65504 saturation to the hardware -- MODE.FP16_OVFL, which a wave only has after
    ``leaf_fp16_sat_mode()``.  A kernel that forgets the call would turn an overflow into inf / NaN activations silently, so the
    convention is linted: the first statement of EVERY ``__global__`` body under leaf_amd/csrc (product and variants) is that call."""
    import glob

    def close(s, i):               # index of the parenthesis that closes the one at s[i]
        d = 0
        while True:
            d += {"(": 1, ")": -1}.get(s[i], 0)
            if d == 0:
                return i
            i += 1
    n, bad = 0, []
    for f in sorted(glob.glob(os.path.join(ROOT, "leaf_amd", "csrc", "**", "*.hip"), recursive=True)):
        s = open(f).read()
        for m in re.finditer(r"__global__", s):
            j = close(s, s.index("(", m.end()))                       # __launch_bounds__(...) or the parameter list
            k = re.match(r"\s*void\s+\w+\s*", s[j + 1:])
            if k:
                j = close(s, s.index("(", j + 1 + k.end() - 1))
            body = s[s.index("{", j) + 1:][:600]
            body = re.sub(r"/\*.*?\*/", "", re.sub(r"//[^\n]*\n", "\n", body), flags=re.S)
            n += 1
            if not body.lstrip().startswith("leaf_fp16_sat_mode();"):
                bad.append((os.path.relpath(f, ROOT), s[:m.start()].count("\n") + 1))
    assert n >= 50, f"only {n} kernels found: the lint's pattern no longer matches the sources"
    assert not bad, f"kernels that do not start with leaf_fp16_sat_mode(): {bad}"


def test_product_has_no_oracle_or_cpu_fallback():
    """The product package must not import the oracle, and must refuse to run without a GPU."""
    import subprocess, sys
    src = "\n".join(open(os.path.join(ROOT, "leaf_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "leaf_amd"))
                    if f.endswith(".py"))
    assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M)
    import torch
    if not torch.cuda.is_available():
        from leaf_amd import _lib
        from leaf_amd.model import LeafCLIPText, get_config
        with pytest.raises(_lib.LeafHipError):
            LeafCLIPText(get_config("tiny-test"), device="cpu")


def test_native_tokenizer_and_stage_match_python(golden_dir):
    """leaf_amd/csrc/host_text.cpp (C++ threads) == the Python tokenizer / mutation on known answers, random printable
    ASCII, and inputs that must take the fallback (non-ASCII, entities)."""
    import random
    import string
    from leaf_amd import attacks
    from leaf_amd.native_text import NativeTokenizer
    from leaf_amd.tokenizer import SimpleTokenizer
    with open(os.path.join(golden_dir, "tokenizer_kat.json")) as f:
        kat = json.load(f)
    nt, pt = NativeTokenizer(n_threads=4), SimpleTokenizer()
    toks, lens = nt.encode_batch_lens(kat["texts"])
    assert np.array_equal(toks, np.array(kat["ids"], dtype=np.int32))
    assert np.array_equal(lens, np.array(kat["ids"]).argmax(-1) + 1)
    rnd = random.Random(0)
    alphabet = string.ascii_letters + string.digits + string.punctuation + "     \t"
    texts = ["".join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 150))) for _ in range(500)]
    texts += ["café au lait", "你好 world", "AT&amp;T &lt;b&gt;", "x <start_of_text> y", "tab\tsep\x1fend", "bell\x07char"]
    assert np.array_equal(nt.encode_batch(texts), pt.encode_batch(texts))
    # fused mutation + tokenisation of a search stage
    sents = ["a photo of a cat", "I'm sure it's 42 degrees, isn't it?", "café au lait", "x", "AT&T tower", "under_score _"]
    rho = 40
    z = np.stack([np.array([rnd.randrange(2 * len(S) + 1) for _ in range(rho)]) for S in sents]).astype(np.int32)
    c = np.array([[rnd.choice(attacks.DEFAULT_V) for _ in range(rho)] for _ in sents], dtype=np.int32)
    c[:, :5] = ord(' ')
    got, lens = nt.mutate_encode(sents, z, c, lambda b, r: attacks._apply_edit(sents[b], int(z[b, r]), int(c[b, r])))
    V = attacks.DEFAULT_V
    want_strings = [attacks.generate_sentence(S, int(z[b, r]), V.index(int(c[b, r])), V, 1, alternative=-1)
                    for b, S in enumerate(sents) for r in range(rho)]
    assert [attacks._apply_edit(S, int(z[b, r]), int(c[b, r])) for b, S in enumerate(sents) for r in range(rho)] == want_strings
    want = pt.encode_batch(want_strings)
    assert np.array_equal(got, want)
    assert np.array_equal(lens, want.argmax(-1) + 1)


def test_persistent_gemm_flush_asm_is_ordered(tmp_path):
    """The persistent GEMM epilogue reads its LDS staging slices with inline-asm ds_read_b128 + counted lgkmcnt waits (a
    compiler-visible LDS read beside in-flight LDS-DMAs would drain vmcnt(0)).  Nothing but program order ties those reads to
    their destination registers, so the generated ISA of every persistent instantiation is checked: each destination is stored
    only behind the wait that releases it and is never copied while in flight (tools/check_flush_asm.py)."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "gemm256h.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-S",
                    "--cuda-device-only", os.path.join(root, "leaf_amd", "csrc", "gemm256h.hip"), "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_flush_asm.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert r.stdout.count("flush blocks") == 8 and " flush blocks 0" not in r.stdout, r.stdout


def test_wgrad_dma_asm_owns_m0(tmp_path):
    """wgrad_tn_dma_kernel issues its LDS-DMA as inline asm (s_mov_b32 m0 + global_load_lds_dwordx4 in ONE statement, M0 cannot
    be declared as clobbered: 'reserved register').  Check in the generated ISA that every DMA is immediately preceded by its own
    M0 write and that nothing else in those kernels reads or writes M0."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "wgrad.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-S",
                    "--cuda-device-only", os.path.join(root, "leaf_amd", "csrc", "wgrad.hip"), "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    lines = out.read_text().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN\w*wgrad_tn_dma_kernel\w*:", l)]
    assert len(starts) == 2
    for s0 in starts:
        end = next(i for i in range(s0, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        body = [l.strip() for l in lines[s0:end] if l.strip() and not l.strip().startswith((";", "."))]
        dma = [i for i, l in enumerate(body) if l.startswith("global_load_lds")]
        assert len(dma) >= 16
        for i in dma:
            assert body[i - 1].startswith("s_mov_b32 m0"), body[i - 1]
        assert sum("m0" in l for l in body) == len(dma)


def test_rng_draws_are_stream_identical():
    """attacks._choice_range replaces np.random.choice(range(N), size, replace) (utils_attacks.py:236,317) by the two primitives numpy's
    legacy RandomState.choice is made of: the numbers AND the state left behind must be identical, or every later draw of a run
    would differ from the reference's."""
    from leaf_amd import attacks
    for seed, (pop, size) in enumerate([(5, 50), (121, 50), (96, 50), (96, 120), (1, 3), (51, 50), (50, 50), (49, 50)]):
        for replace in (size > pop, True) if size <= pop else (True,):
            np.random.seed(seed)
            want = [np.random.choice(range(pop), size=size, replace=replace) for _ in range(3)]
            tail_w = np.random.random(4)
            np.random.seed(seed)
            got = [attacks._choice_range(pop, size, replace) for _ in range(3)]
            tail_g = np.random.random(4)
            assert all(np.array_equal(a, b) for a, b in zip(want, got)) and np.array_equal(tail_w, tail_g), (pop, size, replace)


def test_native_tokenizer_from_concurrent_host_threads():
    """The native pipeline's worker pool runs one job at a time; callers from several Python threads (ctypes releases the GIL during
    the call) must take turns, not corrupt each other's per-worker BPE caches."""
    import threading
    from leaf_amd.native_text import NativeTokenizer
    from leaf_amd.tokenizer import SimpleTokenizer
    tok, ref = NativeTokenizer(n_threads=4), SimpleTokenizer()
    caps = ["a photo of a cat number %d on the table" % i for i in range(120)]
    want = ref.encode_batch(caps)
    bad = []

    def work():
        for _ in range(20):
            if not np.array_equal(tok.encode_batch(caps), want):
                bad.append(1)
    ts = [threading.Thread(target=work) for _ in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


def test_host_text_under_sanitizers(tmp_path):
    """The native host pipeline's text entry points (mutation + BPE, duplicate map, Punkt spans, the three --constrain forms, word
    counts) built with
    AddressSanitizer + UndefinedBehaviorSanitizer on the CPU and fuzzed with random captions, control and non-ASCII bytes included
    (tests/fuzz_host_text.cpp); sanitizers are CPU-only on this pool."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuzz_host_text")
    b = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-fno-omit-frame-pointer", "-pthread", os.path.join(root, "leaf_amd", "csrc", "host_text.cpp"),
                        os.path.join(root, "tests", "fuzz_host_text.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and "sanitize" in b.stderr and "cannot find" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-3000:]
    import gzip
    from leaf_amd.tokenizer import _BPE_PATH
    merges = tmp_path / "merges.txt"
    with gzip.open(_BPE_PATH) as f:
        merges.write_bytes(f.read())
    r = subprocess.run([exe, "4000", str(merges)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz ok" in r.stdout and "runtime error" not in r.stderr, r.stdout[-1500:] + r.stderr[-3000:]


def test_fused_qkv_attention_tile_plan_properties():
    """The host-side cut into M tiles (leaf_qkv_attn_plan, through its test hook): whole sequences, <= tile_rows rows, <= ncap captions."""
    import ctypes as C
    from leaf_amd import _lib
    lib = _lib.diag_lib()           # a diagnostic export (include/leaf_hip_diag.h): tools/diag/libleaf_hip_variants.so
    rng = np.random.default_rng(0)
    for trial in range(20):
        B, rho = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        goff = B if trial % 2 else 0
        lens = np.concatenate([rng.integers(2, 78, goff), rng.integers(1, 78 if trial % 3 else 3, B * rho)]).astype(np.int32)
        out = np.zeros(2 * (lens.size + 2), dtype=np.int32)
        tile_rows, ncap = (256, 3) if trial % 4 < 2 else (128, 1 + trial % 3)
        nt = lib.leaf_debug_qkv_attn_plan(lens.ctypes.data_as(C.c_void_p), 77, 0, lens.size, 1, rho, goff, tile_rows, ncap,
                                          out.ctypes.data_as(C.c_void_p))
        cut, row0 = out[0:2 * (nt + 1):2], out[1:2 * (nt + 1):2]
        assert (row0 == np.concatenate([[0], np.cumsum(lens)])[cut]).all()
        assert cut[0] == 0 and cut[-1] == lens.size and (np.diff(cut) > 0).all()
        for a, b in zip(cut[:-1], cut[1:]):
            assert lens[a:b].sum() <= tile_rows
            caps = [(s - goff) // rho for s in range(a, b) if s >= goff]
            assert not caps or caps[-1] - caps[0] < ncap


def test_native_tokenizer_survives_fork():
    """the persistent worker pool of host_text.cpp is per process -- a forked child (a DataLoader worker, multiprocessing)
    inherits the pool object but none of its threads and used to wait for them for ever; now its first parallel call builds a
    pool of its own, and the parent's keeps working."""
    import multiprocessing as mp
    from leaf_amd.native_text import NativeTokenizer
    texts = [f"a photo of cat number {i} on the wet street" for i in range(400)]
    tok = NativeTokenizer(n_threads=4)
    want = tok.encode_batch(texts)                 # starts the parent's workers

    def child(q):
        q.put(NativeTokenizer(n_threads=4).encode_batch(texts).tobytes() == want.tobytes() and tok.encode_batch(texts).tobytes() == want.tobytes())

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(q,))
    p.start()
    ok = q.get(timeout=60)
    p.join(timeout=30)
    assert ok and p.exitcode == 0
    assert tok.encode_batch(texts).tobytes() == want.tobytes()


def test_configure_runtime_is_explicit_and_warns_when_too_late():
    """importing the package has no process-wide side effect; ``configure_runtime()`` sets HIP_FORCE_DEV_KERNARG=1
    only when the user has not exported it, and says so when the HIP runtime is already up (the setting would be ignored)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, types, warnings
os.environ.pop("HIP_FORCE_DEV_KERNARG", None)
import leaf_amd
assert "HIP_FORCE_DEV_KERNARG" not in os.environ, "import leaf_amd must not touch the environment"
assert "torch" not in sys.modules, "import leaf_amd must not import torch"
r = leaf_amd.configure_runtime()
assert r == {"HIP_FORCE_DEV_KERNARG": "1", "applied": True, "hip_initialised": False}, r
os.environ["HIP_FORCE_DEV_KERNARG"] = "0"
r = leaf_amd.configure_runtime()
assert r["HIP_FORCE_DEV_KERNARG"] == "0" and not r["applied"]          # an exported value wins
del os.environ["HIP_FORCE_DEV_KERNARG"]
fake = types.ModuleType("torch"); fake.cuda = types.SimpleNamespace(is_initialized=lambda: True)
sys.modules["torch"] = fake
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    r = leaf_amd.configure_runtime()
assert r["hip_initialised"] and not r["applied"] and "HIP_FORCE_DEV_KERNARG" not in os.environ
assert len(w) == 1 and "already initialised" in str(w[0].message)
print("ok")
"""
    p = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stdout + p.stderr
