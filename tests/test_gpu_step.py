"""The synthetic-token step used by bench.py (leaf_amd/step.py) on the tiny model: data dependencies and shapes."""
import numpy as np
import pytest

from oracle import text_oracle as O

pytestmark = pytest.mark.gpu


def test_synthetic_step_runs_and_selects_argmax():
    import torch
    from leaf_amd.model import LeafCLIPText, create_model, get_config
    from leaf_amd.step import StepConfig, SyntheticCandidates, search_synthetic, train_step_tokens
    m = create_model("tiny-test-quickgelu", seed=12, trainable=True)
    frozen = LeafCLIPText(get_config("tiny-test-quickgelu")).copy_from(m)
    base = torch.from_numpy(O.synthetic_tokens(8, seed=4).astype(np.int32)).cuda()
    sc = StepConfig(rho=10, k_adv=2, lr=1e-4)
    anchor = frozen.encode_text(base)
    adv = search_synthetic(m, anchor, base, sc, seed=0)
    assert adv.shape == base.shape and adv.dtype == torch.int32
    diff = (adv != base).sum(-1).cpu().numpy()
    assert diff.max() <= sc.k_adv and (adv.argmax(-1) == base.argmax(-1)).all()   # <= k edits, EOT untouched
    # candidates differ from the current row in exactly <= 1 position, inside the caption
    gen = SyntheticCandidates(base, None, 10, m.cfg.vocab_size, 0)
    cand, pos = gen.stage1(base)
    d = (cand != base[:, None, :])
    assert d.sum(-1).max() <= 1
    eot = base.argmax(-1).cpu().numpy()
    assert (pos >= 1).all() and (pos < eot[:, None]).all()
    # prefix reuse and plain EOT trimming pick the same adversarial ids (bit-identical scoring)
    lens = eot + 1
    a1 = search_synthetic(m, anchor, base, sc, seed=3, base_lens=lens, prefix_reuse=True)
    a2 = search_synthetic(m, anchor, base, sc, seed=3, base_lens=lens, prefix_reuse=False)
    assert torch.equal(a1, a2)
    l0 = float(train_step_tokens(m, frozen, base, sc, seed=1))
    l1 = float(train_step_tokens(m, frozen, base, sc, seed=1))
    assert np.isfinite([l0, l1]).all()
