#!/usr/bin/env python3
"""LEAF text-encoder adversarial fine-tuning on MI355X -- command-line drop-in for the reference's
``train_AT_text_only.py`` (same flags: leaf_amd/params.py; same experiment folder, ``results.csv`` and
``epoch_latest.pt`` layout: train_AT_text_only.py:57,483,516-525).

    python3 train_AT_text_only.py --model hf-hub:chs20/fare2-clip --pretrained /path/to/open_clip_pytorch_model.bin \
        --train-data 'shards/{00000000..00001287}.tar' --dataset-type webdataset --train-num-samples 80000 \
        --batch-size 128 --lr 1e-5 --wd 1e-4 --warmup 1400 --epochs 30 --k_adv 1 --rho 50 --constrain --seed 1
    python3 -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -- train_AT_text_only.py ...   # RCCL DP ("--": --logs is an ambiguous prefix of torchrun's own --logs-specs)

Differences from the reference, all outside the hot path: downstream evaluation (ImageNet / AG-News zero-shot,
utils_AT.py:428-556) is not run; ``hf-hub:`` ids select the architecture but weights come from ``--pretrained``
(no network); with WORLD_SIZE > 1 the step really is data parallel (the reference's DDP wrapper cannot reach
``encode_text``).
"""
import logging
import os

from leaf_amd import configure_runtime

configure_runtime()   # HIP_FORCE_DEV_KERNARG=1, before torch loads the HIP runtime (leaf_amd/__init__.py says why)
import random
import string
import sys
from datetime import datetime

import numpy as np
import torch

from leaf_amd.params import parse_args
from leaf_amd.tokenizer import get_tokenizer
from leaf_amd.train import (LATEST_CHECKPOINT_NAME, LeafAdamW, get_latest_checkpoint, get_text_data, is_master, load_checkpoint,
                            make_scheduler, save_checkpoint, train_one_epoch_text_only)


def random_seed(seed=42, rank=0):
    torch.manual_seed(seed + rank)
    np.random.seed(seed + rank)
    random.seed(seed + rank)


def init_distributed_device(args):
    args.distributed, args.world_size, args.rank, args.local_rank = False, 1, 0, 0
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        args.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        ndev = torch.cuda.device_count()
        if args.dist_backend == "gloo" and ndev and args.local_rank >= ndev:
            # rehearsal on a box with fewer GPUs than ranks (gloo moves CUDA tensors through the host; RCCL wants a device per rank)
            args.local_rank %= ndev
        torch.cuda.set_device(args.local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend=args.dist_backend)
        args.world_size, args.rank, args.distributed = torch.distributed.get_world_size(), torch.distributed.get_rank(), True
    if not torch.cuda.is_available():
        raise SystemExit("train_AT_text_only.py needs an MI355X: the HIP engine has no CPU path")
    args.device = f"cuda:{args.local_rank}"
    torch.cuda.set_device(args.device)
    return torch.device(args.device)


def main(argv):
    args = parse_args(argv)
    if args.lock_text or args.lock_text_unlocked_layers or args.distill_model:
        # accepted by the parser for command-line compatibility, but they would change WHAT is trained: refuse rather than ignore
        raise SystemExit("--lock-text / --lock-text-unlocked-layers / --distill-model are not supported by the text-only engine "
                         "(the reference's CLIP class has no lock_text_tower either: src/open_clip/model.py:256-262)")
    V = [-1] + [ord(c) for c in string.ascii_lowercase + ' ' + string.ascii_uppercase + string.digits + string.punctuation]
    device = init_distributed_device(args)
    if args.name is None:
        safe = args.model.replace('/', '-').replace(':', '-')
        date_str = datetime.now().strftime("%Y_%m_%d-%H_%M_%S")
        if args.distributed:
            obj = [date_str]
            torch.distributed.broadcast_object_list(obj, src=0)
            date_str = obj[0]
        args.name = '-'.join([date_str, f"model_{safe}", f"lr_{args.lr}", f"b_{args.batch_size}", f"f_{args.accum_freq}",
                              f"k_{args.k_adv}", f"rho{args.rho}"])
    log_base = os.path.join(args.logs, args.name)
    args.checkpoint_path = os.path.join(log_base, "checkpoints")
    handlers = [logging.StreamHandler()]
    exists = [False]
    if is_master(args):
        os.makedirs(args.checkpoint_path, exist_ok=True)
        log_path = os.path.join(log_base, "out.log")
        exists[0] = os.path.exists(log_path) and args.resume != "latest"
    if args.distributed:
        # the master alone looks at the folder (train_AT_text_only.py:133-137), but EVERY rank has to leave: one that walked on
        # would wait in the next collective for ever
        torch.distributed.broadcast_object_list(exists, src=0)
    if exists[0]:
        if is_master(args):
            print("Error. Experiment already exists. Use --name {} to specify a new experiment.")
        if args.distributed:
            torch.distributed.destroy_process_group()
        return -1
    if is_master(args):
        handlers.append(logging.FileHandler(log_path))
    logging.basicConfig(level=logging.INFO, format="%(asctime)s | %(levelname)s | %(message)s", handlers=handlers)
    out_dir = f'./results/{args.custom_out_folder}text_only_k{args.k_adv}_rho{args.rho}_seed{args.seed}'
    if args.resume == "latest":
        # the reference looks under logs/<name>/checkpoints, where it never writes (its checkpoints go to ./results/...,
        # train_AT_text_only.py:483,560-569); look in both places so that --resume latest really resumes
        args.resume = get_latest_checkpoint(args.checkpoint_path) or get_latest_checkpoint(out_dir)
        if args.resume is None:
            logging.warning(f"--resume latest: no checkpoint under {args.checkpoint_path} or {out_dir}; starting from epoch 0")

    from leaf_amd.model import LeafCLIPText, create_model
    dtype = "bf16" if "bf16" in args.precision or "bfloat16" in args.precision else "fp16"
    if args.precision == "fp32":
        logging.warning("--precision fp32 is not offered by the MFMA path; using fp16 operands with fp32 accumulation")
    name = args.model + ("-quickgelu" if args.force_quick_gelu and not args.model.endswith("quickgelu") and not args.model.startswith("hf-hub:") else "")
    if not args.pretrained and not args.resume and not args.random_init and not name.startswith("tiny-test"):
        # the reference would download the hub weights; training a randomly initialised tower against a random frozen copy of
        # itself is meaningless, so it has to be asked for
        raise SystemExit(f"--pretrained is empty: pass a local checkpoint for '{args.model}' (OpenCLIP .bin/.pt, HF safetensors or a "
                         "directory; there is no network here), or --random-init to train from seeded random weights on purpose")
    if not args.pretrained and is_master(args):
        logging.warning("no --pretrained checkpoint: the text tower (and its frozen TextFARE anchor) start from seeded RANDOM weights")
    model = create_model(name, device=device, dtype=dtype, pretrained=args.pretrained or None, trainable=True, seed=args.seed)
    if getattr(args, "arithmetic", None):
        model.set_precision(args.arithmetic)
    random_seed(args.seed, args.rank)
    if is_master(args):
        with open(os.path.join(log_base, "params.txt"), "w") as f:
            for k in sorted(vars(args)):
                f.write(f"{k}: {getattr(args, k)}\n")
    optimizer = LeafAdamW(model, lr=args.lr, betas=(args.beta1, args.beta2), eps=args.eps, weight_decay=args.wd,
                          lock_image=args.lock_image)
    start_epoch = 0
    if args.resume:
        start_epoch = load_checkpoint(args.resume, model, optimizer)
        logging.info(f"=> resuming checkpoint '{args.resume}' (epoch {start_epoch})")
    tokenizer = get_tokenizer(args.model)
    if args.constrain:
        from leaf_amd import attacks
        if args.punkt_params and not (args.dictionary_file and args.dictionary_tokenizer == "treebank"):
            raise SystemExit("--punkt-params needs --dictionary-file and --dictionary-tokenizer treebank")
        attacks.set_dictionary(attacks.Dictionary.from_file(args.dictionary_file, tokenizer=args.dictionary_tokenizer,
                                                            punkt_params=args.punkt_params) if args.dictionary_file
                               else attacks.Dictionary.from_nltk())
    data = get_text_data(args, epoch=start_epoch)
    total_steps = (data["train"].dataloader.num_batches // args.accum_freq) * args.epochs
    scheduler = make_scheduler(args, optimizer, total_steps, data["train"].dataloader.num_batches)
    # frozen anchor model = the weights the run STARTED from (train_AT_text_only.py:439-465)
    frozen = LeafCLIPText(model.cfg, device=device, dtype=dtype)
    if args.resume and args.pretrained:
        from leaf_amd.checkpoint import load_checkpoint_file
        frozen.load_state_dict(load_checkpoint_file(args.pretrained))
        frozen.set_split_masks(model.split_masks)
    else:
        if args.resume and is_master(args):
            # the reference re-creates model_frozen from the hub weights on every start (train_AT_text_only.py:439-465); without
            # --pretrained there is nothing to restore the anchor from but the resumed, already adversarially trained weights
            logging.warning("--resume WITHOUT --pretrained: the frozen TextFARE anchor is a copy of the RESUMED weights, not of the "
                            "weights the run started from -- the objective differs from the interrupted run's; pass the original "
                            "--pretrained checkpoint to continue it unchanged")
        frozen.copy_from(model)
    frozen.pack()
    model.pack()
    if is_master(args):
        os.makedirs(out_dir, exist_ok=True)
    results = []
    if args.resume and is_master(args):   # train_AT_text_only.py:371-372: results.csv beside the checkpoint continues
        prev = os.path.join(os.path.abspath(os.path.join(args.resume, os.pardir)), "results.csv")
        if os.path.exists(prev):
            import pandas as pd
            results = [r for r in pd.read_csv(prev).to_dict(orient="records") if r.get("epoch", 0) <= start_epoch]
            logging.info(f"=> {len(results)} earlier epochs of {prev} carried over")
    for epoch in range(start_epoch, args.epochs):
        if is_master(args):
            logging.info(f'Start epoch {epoch}')
        log = train_one_epoch_text_only(model, frozen, tokenizer, V, data, None, epoch, optimizer, None, scheduler, args)
        completed = epoch + 1
        if args.distributed:
            # every rank applied the same summed gradient to the same weights: they must still be the same bits (one small
            # all-gather per epoch; a replica that drifted would otherwise only show up as a slowly diverging loss)
            chk = [None] * args.world_size
            torch.distributed.all_gather_object(chk, (float(model.flat.double().sum()), float(model.flat.double().abs().sum())))
            if any(c != chk[0] for c in chk):
                raise RuntimeError(f"epoch {completed}: replicas differ across ranks (weight checksums {chk})")
            if is_master(args):
                logging.info(f"epoch {completed}: weights identical on {args.world_size} ranks (checksum {chk[0][0]:.9g})")
        if is_master(args):
            results.append({"epoch": completed, **{k.replace("train/", ""): v for k, v in log.items()}})
            import pandas as pd
            pd.DataFrame(results).to_csv(os.path.join(out_dir, "results.csv"), index=False)
            if completed == args.epochs or (args.save_frequency > 0 and completed % args.save_frequency == 0):
                save_checkpoint(os.path.join(out_dir, LATEST_CHECKPOINT_NAME), completed, args.name, model, optimizer)
            if completed == args.epochs and args.export_hf:
                # release format of the reference (README.md:98, conversion/convert_2.py): HF CLIPTextModel(WithProjection)
                from leaf_amd.checkpoint import write_hf_text_model
                write_hf_text_model(args.export_hf, model.state_dict(), model.cfg, with_projection=True)
                logging.info(f"=> HuggingFace CLIPTextModelWithProjection written to {args.export_hf}")
    if args.distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]) or 0)
