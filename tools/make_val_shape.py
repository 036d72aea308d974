"""Writes fastspeech2_amd/workloads/ljspeech_val_phonemes.json: the phoneme count of every line of the reference's
preprocessed_data/LJSpeech/val.txt IN FILE ORDER (what synthesize.py --mode batch feeds, 8 lines per batch, unsorted:
synthesize.py:197-199, dataset.py:150-198).  Runs in the build container only (/root/reference is not on the GPU box);
the table it writes is committed and is all bench.py needs to shape the batch-synthesis workload like val.txt."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastspeech2_amd.text import text_to_sequence  # noqa: E402

REF = os.environ.get("FS2_REFERENCE", "/root/reference")
src = os.path.join(REF, "preprocessed_data", "LJSpeech", "val.txt")
counts = []
with open(src, encoding="utf-8") as f:
    for line in f:
        n, s, t, r = line.strip("\n").split("|")
        counts.append(len(text_to_sequence(t, ["english_cleaners"])))
out = os.path.join(ROOT, "fastspeech2_amd", "workloads", "ljspeech_val_phonemes.json")
hist = {}
for c in counts:
    hist[c] = hist.get(c, 0) + 1
json.dump({"source": "preprocessed_data/LJSpeech/val.txt (reference), phoneme ids per line via text_to_sequence, file order",
           "n": len(counts), "min": min(counts), "max": max(counts), "mean": round(sum(counts) / len(counts), 2),
           "histogram": {str(k): hist[k] for k in sorted(hist)}, "counts": counts}, open(out, "w"))
print(len(counts), min(counts), max(counts), sum(counts) / len(counts))
