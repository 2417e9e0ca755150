#!/bin/bash
# End-to-end throughput of the REAL trainer (strings in, train_one_epoch_text_only): ViT-L, B = 128, rho = 50, k = 1, synthetic captions,
# with and without --constrain (word-list file + the regex word tokenizer).  Prints the trainer's own samples/s log lines.
# usage: tools/trainer_bench.sh OUTDIR
OUT=$(realpath -m $1); mkdir -p $OUT
W=/tmp/leaf_trainer_bench; rm -rf $W; mkdir -p $W      # checkpoints and the word list stay out of OUT (only logs go there)
python - <<'PY' > $W/words.txt
import random
from leaf_amd.train import _SYN_WORDS
rng = random.Random(0)
w = set(_SYN_WORDS)
while len(w) < 236736:
    w.add("".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(2, 10))))
print("\n".join(sorted(w)))
PY
ROOT=$PWD
for c in "" "--constrain --dictionary-file $W/words.txt"; do
  name=run$( [ -n "$c" ] && echo _constrain )
  (cd $W && timeout -k 10 400 python $ROOT/train_AT_text_only.py --model ViT-L-14-quickgelu --random-init --dataset-type synthetic \
     --train-num-samples 7680 --batch-size 128 --epochs 1 --rho 50 --k_adv 1 --lr 1e-5 --wd 1e-4 --warmup 10 --log-every-n-steps 10 \
     --save-frequency 0 --seed 1 --custom_out_folder b_ --logs $W/logs --name $name $c > $OUT/$name.log 2>&1)
  echo "== ${c:-unconstrained}"; grep -o "Batch (t): [0-9.]*, [0-9.]*/s" $OUT/$name.log | tail -4
done
