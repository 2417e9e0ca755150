"""Command-line flags of ``train_AT_text_only.py`` (drop-in for the reference's ``params_AT.py:38-606``).

Every flag the reference's LEAF launch scripts pass (scripts/train_leaf_vit*.sh) is accepted with the same name, type
and default.  Flags that only concern parts of the reference outside the text hot path (image tower, evaluation
sets, wandb, remote sync ...) are parsed and ignored, so an existing command line keeps working; the ones that
would change the text path's behaviour in a way this engine does not implement are rejected loudly
(``--use_charmer``, ``--horovod``); ``--normalize_fare`` is implemented in the training kernels.

Model-dependent Adam defaults follow ``get_default_params`` (params_AT.py:17-23): "vit" in the lower-cased model
name -> beta2 0.98 / eps 1e-6, else 0.999 / 1e-8 (so ``hf-hub:chs20/fare2-clip`` gets the latter, as in the
reference).
"""
import argparse


def str2float(x):
    if '/' in x:
        n, d = x.split('/')
        return float(n) / float(d)
    try:
        return float(x)
    except Exception:
        raise argparse.ArgumentTypeError('Fraction or float value expected.')


def get_default_params(model_name):
    if "vit" in model_name.lower():
        return {"lr": 5.0e-4, "beta1": 0.9, "beta2": 0.98, "eps": 1.0e-6}
    return {"lr": 5.0e-4, "beta1": 0.9, "beta2": 0.999, "eps": 1.0e-8}


# (flag, kwargs) accepted for command-line compatibility but unused by the text path
_IGNORED = [
    ("--train-data-upsampling-factors", dict(type=str, default=None)), ("--val-data", dict(type=str, default=None)),
    ("--val-text-classification", dict(type=str, default='fancyzhx/ag_news')), ("--val-num-samples", dict(type=int, default=None)),
    ("--dataset-resampled", dict(action="store_true")), ("--csv-separator", dict(type=str, default="\t")),
    ("--csv-img-key", dict(type=str, default="filepath")), ("--imagenet-val", dict(type=str, default=None)),
    ("--imagenet-v2", dict(type=str, default=None)), ("--log-local", dict(action="store_true")),
    ("--epochs-cooldown", dict(type=int, default=None)), ("--use-bn-sync", dict(action="store_true")),
    ("--lr-cooldown-end", dict(type=float, default=0.0)), ("--lr-cooldown-power", dict(type=float, default=1.0)),
    ("--save-most-recent", dict(action="store_true")), ("--zeroshot-frequency", dict(type=int, default=2)),
    ("--val-frequency", dict(type=int, default=1)), ("--pretrained-image", dict(action="store_true")),
    ("--lock-image", dict(action="store_true")), ("--lock-image-unlocked-groups", dict(type=int, default=0)),
    ("--lock-image-freeze-bn-stats", dict(action="store_true")), ("--grad-checkpointing", dict(action="store_true")),
    ("--local-loss", dict(action="store_true")), ("--gather-with-grad", dict(action="store_true")),
    ("--force-patch-dropout", dict(type=float, default=None)), ("--force-custom-text", dict(action="store_true")),
    ("--torchscript", dict(action="store_true")), ("--torchcompile", dict(action="store_true")),
    ("--trace", dict(action="store_true")), ("--dist-url", dict(type=str, default="env://")),
    ("--report-to", dict(type=str, default='')), ("--wandb-notes", dict(type=str, default='')),
    ("--wandb-project-name", dict(type=str, default='open-clip')), ("--debug", dict(action="store_true")),
    ("--copy-codebase", dict(action="store_true")), ("--ddp-static-graph", dict(action="store_true")),
    ("--no-set-device-rank", dict(action="store_true")), ("--lock-text", dict(action="store_true")),
    ("--lock-text-unlocked-layers", dict(type=int, default=0)), ("--lock-text-freeze-layer-norm", dict(action="store_true")),
    ("--coca-caption-loss-weight", dict(type=float, default=2.0)), ("--coca-contrastive-loss-weight", dict(type=float, default=1.0)),
    ("--remote-sync", dict(type=str, default=None)), ("--remote-sync-frequency", dict(type=int, default=300)),
    ("--remote-sync-protocol", dict(choices=["s3", "fsspec"], default="s3")), ("--delete-previous-checkpoint", dict(action="store_true")),
    ("--distill-model", dict(default=None)), ("--distill-pretrained", dict(default=None)), ("--use-bnb-linear", dict(default=None)),
    ("--siglip", dict(action="store_true")), ("--eps_adv", dict(type=str2float, default='2/255')),
    ("--stepsize_adv", dict(type=str2float, default=None)), ("--n_steps_adv", dict(type=int, default=10)),
    ("--k_adv_test", dict(type=int, default=1)), ("--n_charmer_test", dict(type=int, default=20)),
    ("--n_val_imagenet", dict(type=int, default=1000)), ("--w_contrastive", dict(type=float, default=1)),
    ("--w_fare_text", dict(type=float, default=0)), ("--w_fare_image", dict(type=float, default=0)),
    ("--attack_objective", dict(type=str, default='fare')), ("--text_only", dict(action='store_true')),
    ("--n_val_text", dict(type=int, default=200)),
]


def parse_args(args):
    p = argparse.ArgumentParser()
    p.add_argument("--train-data", type=str, default=None,
                   help="webdataset tar shards (brace pattern), a text file with one caption per line, or a csv")
    p.add_argument("--train-num-samples", type=int, default=None)
    p.add_argument("--dataset-type", choices=["webdataset", "csv", "synthetic", "text", "auto"], default="auto")
    p.add_argument("--csv-caption-key", type=str, default="title")
    p.add_argument("--logs", type=str, default="./logs/")
    p.add_argument("--name", type=str, default=None)
    p.add_argument("--workers", type=int, default=4)
    p.add_argument("--batch-size", type=int, default=64)
    p.add_argument("--epochs", type=int, default=32)
    p.add_argument("--lr", type=float, default=None)
    p.add_argument("--beta1", type=float, default=None)
    p.add_argument("--beta2", type=float, default=None)
    p.add_argument("--eps", type=float, default=None)
    p.add_argument("--wd", type=float, default=0.2)
    p.add_argument("--warmup", type=int, default=10000)
    p.add_argument("--skip-scheduler", action="store_true", default=False)
    p.add_argument("--lr-scheduler", type=str, default='cosine')
    p.add_argument("--save-frequency", type=int, default=1)
    p.add_argument("--resume", default=None, type=str)
    p.add_argument("--precision", choices=["amp", "amp_bf16", "amp_bfloat16", "bf16", "fp16", "pure_bf16", "pure_fp16", "fp32"],
                   default="amp", help="amp / fp16 -> fp16 MFMA operands (default, like the reference's fp16 autocast); "
                                       "*bf16 -> bf16 operands; fp32 is not offered by the MFMA path")
    p.add_argument("--model", type=str, default="RN50")
    p.add_argument("--pretrained", default='', type=str, help="local checkpoint (OpenCLIP .bin/.pt, HF safetensors or a directory)")
    p.add_argument("--random-init", default=False, action='store_true',
                   help="train from seeded random weights on purpose (without it an empty --pretrained is an error)")
    p.add_argument("--export-hf", type=str, default=None,
                   help="after the last epoch also write the text tower as a HuggingFace CLIPTextModelWithProjection "
                        "(model.safetensors + config.json) into this directory")
    p.add_argument("--force-quick-gelu", default=False, action='store_true')
    p.add_argument("--accum-freq", type=int, default=1)
    p.add_argument("--dist-backend", default="nccl", type=str)
    p.add_argument("--horovod", default=False, action="store_true")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--grad-clip-norm", type=float, default=None)
    p.add_argument("--log-every-n-steps", type=int, default=1)
    p.add_argument("--use_charmer", default=False, action="store_true")
    p.add_argument("--k_adv", type=int, default=1)
    p.add_argument("--rho", type=int, default=20)
    p.add_argument("--constrain", default=False, action="store_true")
    p.add_argument("--dictionary-file", type=str, default=None,
                   help="word list (one per line) for --constrain when the nltk corpus is unavailable")
    p.add_argument("--dictionary-tokenizer", type=str, default="regex", choices=["regex", "treebank"],
                   help="with --dictionary-file: 'treebank' = nltk.word_tokenize restated (leaf_amd/treebank.py); 'regex' = a plain "
                        "letters/digits splitter")
    p.add_argument("--punkt-params", type=str, default=None,
                   help="with --dictionary-tokenizer treebank: nltk's trained Punkt tables exported by tools/export_punkt_params.py, "
                        "so that sentences are split as nltk.word_tokenize splits them, without nltk")
    p.add_argument("--normalize_fare", default=False, action='store_true')
    p.add_argument("--precise-anchor", default=False, action="store_true",
                   help="(not in the reference) frozen model's anchor pass in fp32-grade arithmetic (leaf_text_forward_precise); default: "
                        "the 16-bit arithmetic the candidates are scored in, as the reference computes both sides in one arithmetic")
    p.add_argument("--arithmetic", default=None, choices=["rowsafe", "fast"],
                   help="(not in the reference) arithmetic of the forward-only passes: rowsafe (default) keeps every embedding row within "
                        "1e-3 of fp32; fast drops the split GEMMs (-4 %% step time, 1 %% of the rows above 1e-3)")
    p.add_argument("--custom_out_folder", type=str, default='')
    for flag, kw in _IGNORED:
        p.add_argument(flag, **kw)
    a = p.parse_args(args)
    for name, val in get_default_params(a.model).items():
        if getattr(a, name) is None:
            setattr(a, name, val)
    return a
