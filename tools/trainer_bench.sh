#!/bin/bash
# End-to-end throughput of the REAL trainer (strings in, train_one_epoch_text_only): ViT-L, B = 128, rho = 50, k = 1, synthetic captions,
# with and without --constrain (word-list file + the regex word tokenizer), and with --constrain on web-style captions from a file
# (punctuation, 15 % two-sentence) under the reference's word tokenizer restated (Treebank step + Punkt tables).  Prints the trainer's
# own samples/s log lines.
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
# web-style captions for the third run: punctuation, some with two sentences ("vintage chair. free shipping!"), and a small Punkt
# table set in the format of tools/export_punkt_params.py
python - <<'PY'
import json, random
from leaf_amd.train import _SYN_WORDS
rng = random.Random(1)
caps = []
for _ in range(7680):
    w = [rng.choice(_SYN_WORDS) for _ in range(rng.randint(3, 14))]
    for _ in range(rng.randint(0, 2)):
        i = rng.randrange(len(w))
        w[i] = rng.choice([w[i] + ",", w[i] + "'s", '"' + w[i] + '"', "(" + w[i] + ")", w[i] + "!", w[i] + ":", "dr. " + w[i], "no. 5 " + w[i]])
    if len(w) > 3 and rng.random() < 0.15:
        w[rng.randrange(1, len(w) - 1)] += "."
    caps.append(" ".join(w) + rng.choice([".", "", ".", "!"]))
open("/tmp/leaf_trainer_bench/captions.txt", "w").write("\n".join(caps))
json.dump({"abbrev_types": ["dr", "mr", "mrs", "st", "e.g", "i.e", "vs", "inc", "co", "no", "p.m", "a.m"], "collocations": [["##number##", "street"]],
           "sent_starters": ["the", "a"], "ortho_context": {"the": 34, "a": 50}}, open("/tmp/leaf_trainer_bench/punkt.json", "w"))
PY
ROOT=$PWD
# fourth run: the same captions as DataComp-style webdataset shards (per sample a ~60-KB dummy .jpg, the .txt, a .json), read by the
# header-scanning tar reader on the loader's background thread: "Load (t)" in the log line is what the step waited for a batch
python - <<'PY'
import sys
sys.path.insert(0, "tools")
import io, os, random, tarfile
caps = open("/tmp/leaf_trainer_bench/captions.txt").read().split("\n")
os.makedirs("/tmp/leaf_trainer_bench/shards", exist_ok=True)
blob = os.urandom(60 * 1024)
per = 1920
for s in range(len(caps) // per):
    with tarfile.open(f"/tmp/leaf_trainer_bench/shards/{s:08d}.tar", "w", format=tarfile.USTAR_FORMAT) as tf:
        for i, c in enumerate(caps[s * per:(s + 1) * per]):
            for ext, data in ((".jpg", blob), (".txt", c.encode()), (".json", b'{"width": 512}')):
                ti = tarfile.TarInfo(f"{s:05d}{i:05d}{ext}"); ti.size = len(data); tf.addfile(ti, io.BytesIO(data))
PY
i=0
for c in "" "--constrain --dictionary-file $W/words.txt" \
         "--constrain --dictionary-file $W/words.txt --dictionary-tokenizer treebank --punkt-params $W/punkt.json --dataset-type text --train-data $W/captions.txt" \
         "--constrain --dictionary-file $W/words.txt --dictionary-tokenizer treebank --punkt-params $W/punkt.json --dataset-type webdataset --train-data $W/shards/{00000000..00000003}.tar --workers 8"; do
  i=$((i + 1)); name=run$i
  (cd $W && timeout -k 10 400 python $ROOT/train_AT_text_only.py --model ViT-L-14-quickgelu --random-init --dataset-type synthetic \
     --train-num-samples 7680 --batch-size 128 --epochs 1 --rho 50 --k_adv 1 --lr 1e-5 --wd 1e-4 --warmup 10 --log-every-n-steps 10 \
     --save-frequency 0 --seed 1 --custom_out_folder b_ --logs $W/logs --name $name $c > $OUT/$name.log 2>&1)
  echo "== ${c:-unconstrained}"; grep -o "Data (t): [0-9.]* Batch (t): [0-9.]*, [0-9.]*/s\|Load (t): [0-9.]*" $OUT/$name.log | paste - - | tail -4
done
