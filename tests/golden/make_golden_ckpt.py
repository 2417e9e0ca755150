#!/usr/bin/env python3
"""ckpt_structure.json: what the REFERENCE writes into ``epoch_latest.pt`` -- produced by EXECUTING the reference's own
statements in the reference's own order.

The statements are taken out of /root/reference/train_AT_text_only.py by their text (nothing is restated or re-ordered here):

  (L) the ``--lock-image`` block                         (:286-290)   only in the ``lock_image`` case
  (O) the AdamW construction over ``model.named_parameters()``   (:326-341)
  (F) the freeze of ``model.visual``                     (:489-490)
  (C) the checkpoint dict                                (:518-523)

and run in the order of their line numbers, which this script asserts: L < O < F < C.  In particular the optimizer is built
BEFORE the image tower is frozen, so (without ``--lock-image``) its two groups contain every ``visual.*`` parameter -- round 2's
fixture reversed O and F and pinned a layout the reference never writes.  Between F and C one training step
of the text path runs (``encode_text`` -> loss -> backward -> ``optimizer.step()``), so the text parameters carry AdamW state
and the frozen / unused ones do not.

Only structure is stored (names, shapes, ids, hyper-parameters): no tensor data, no reference source.
Runs only in the build container.   python tests/golden/make_golden_ckpt.py
"""
import argparse
import json
import os
import re
import sys
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

REF_MAIN = os.path.join(MG.REF, "train_AT_text_only.py")


def ref_block(src_lines, first_pat, last_pat):
    """(first line number, dedented source) of the statement block that starts at the first line matching ``first_pat`` and ends
    with the first later line matching ``last_pat`` (inclusive)."""
    i0 = next(i for i, l in enumerate(src_lines) if re.search(first_pat, l))
    i1 = next(i for i in range(i0, len(src_lines)) if re.search(last_pat, src_lines[i]))
    return i0 + 1, textwrap.dedent("".join(src_lines[i0:i1 + 1]))


def run_case(lock_image: bool):
    from open_clip.model import CLIP
    from torch import optim
    from oracle import text_oracle as O
    src = open(REF_MAIN).read().splitlines(keepends=True)
    ln_lock, blk_lock = ref_block(src, r"^\s*if args\.lock_image:", r"freeze_bn_stats=args\.lock_image_freeze_bn_stats\)")
    ln_opt, blk_opt = ref_block(src, r"^\s*exclude = lambda n, p:", r"^\s*eps=args\.eps,\s*$")
    blk_opt += ")\n"                                          # the closing parenthesis of optim.AdamW( ... on the following line
    ln_frz, blk_frz = ref_block(src, r"^\s*for param in model\.visual\.parameters\(\):", r"param\.requires_grad_\(False\)")
    ln_ck, blk_ck = ref_block(src, r"^\s*checkpoint_dict = \{", r"^\s*\}\s*$")
    assert ln_lock < ln_opt < ln_frz < ln_ck, "the reference's own order: lock-image, optimizer, freeze, checkpoint"

    torch.manual_seed(0)
    model = CLIP(**MG.TINY, quick_gelu=True).float().train()
    args = argparse.Namespace(lock_image=lock_image, lock_image_unlocked_groups=0, lock_image_freeze_bn_stats=False,
                              wd=1e-4, lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-8, name="fixture")
    ns = {"model": model, "original_model": model, "args": args, "optim": optim, "completed_epoch": 1}
    exec(blk_lock, ns)                                        # (L)
    visual_trainable_at_build = any(p.requires_grad for p in model.visual.parameters())
    exec(blk_opt, ns)                                         # (O)
    optimizer = ns["optimizer"]
    exec(blk_frz, ns)                                         # (F)
    toks = torch.from_numpy(O.synthetic_tokens(4, seed=2).astype(np.int64))
    f = model.encode_text(toks)
    (f ** 2).sum().backward()
    optimizer.step()
    exec(blk_ck, ns)                                          # (C)
    ck = ns["checkpoint_dict"]
    assert sorted(ck) == ["epoch", "name", "optimizer", "state_dict"]
    osd = ck["optimizer"]
    named = list(model.named_parameters())
    by_obj = {id(p): n for n, p in named}
    group_names = [[by_obj[id(p)] for p in g["params"]] for g in optimizer.param_groups]
    first = next(iter(osd["state"].values()))
    return {
        "reference_lines": {"lock_image": ln_lock, "optimizer": ln_opt, "freeze_visual": ln_frz, "checkpoint_dict": ln_ck},
        "visual_trainable_when_optimizer_was_built": visual_trainable_at_build,
        "checkpoint_keys": sorted(ck),
        "group_names": group_names,
        "param_groups": [{k: (list(v) if k in ("params", "betas") else v) for k, v in g.items()} for g in osd["param_groups"]],
        "state_ids": sorted(osd["state"].keys()),
        "state_names": [[n for g in group_names for n in g][i] for i in sorted(osd["state"].keys())],
        "state_entry_keys": sorted(first.keys()),
        "step_dtype": str(first["step"].dtype), "step_shape": list(first["step"].shape), "step_value": float(first["step"]),
        "no_state_names": [n for i, n in enumerate(n for g in group_names for n in g) if i not in osd["state"]],
    }


def main():
    MG.install_stubs()
    from open_clip.model import CLIP
    torch.set_num_threads(4)
    model = CLIP(**MG.TINY, quick_gelu=True).float()
    struct = {
        "generator": "tests/golden/make_golden_ckpt.py: the reference's own statements, executed in the reference's order",
        "state_dict_keys": [[k, list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()],
        "named_parameters": [[n, p.ndim] for n, p in model.named_parameters()],
        "cases": {"default": run_case(False), "lock_image": run_case(True)},
    }

    def jsonable(o):
        if isinstance(o, dict):
            return {str(k): jsonable(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [jsonable(v) for v in o]
        if isinstance(o, (bool, int, float, str)) or o is None:
            return o
        return str(o)
    with open(os.path.join(HERE, "ckpt_structure.json"), "w") as fjs:
        json.dump(jsonable(struct), fjs)
    for name, c in struct["cases"].items():
        print(name, "groups", [len(g) for g in c["group_names"]], "states", len(c["state_ids"]),
              "visual trainable at optimizer build:", c["visual_trainable_when_optimizer_was_built"])


if __name__ == "__main__":
    main()
