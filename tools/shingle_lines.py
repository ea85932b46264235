"""Which lines of a repository source carry 8-token shingles that also occur in the reference (see shingle_overlap.py)."""
import re
import sys
from collections import Counter
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import shingle_overlap as S

ref_files = [p for p in S.REF.rglob("*") if p.suffix in (".cpp", ".h", ".cl", ".frag", ".hpp") and "ext/clogs" not in str(p)]
every = set()
for p in ref_files:
    every |= S.shingles(p)
for f in sys.argv[1:]:
    text = Path(f).read_text()
    text = re.sub(r"/\*(?:.|\n)*?\*/", lambda m: "\n" * m.group(0).count("\n"), text)
    toks, ln = [], []
    for i, l in enumerate(text.split("\n"), 1):
        l = re.sub(r"//.*", "", l)
        for t in S.TOKEN.findall(l):
            toks.append(t)
            ln.append(i)
    c = Counter()
    for i in range(len(toks) - 7):
        if tuple(toks[i:i + 8]) in every:
            c[ln[i]] += 1
    total = sum(c.values())
    print(f, total, "shared shingles of", len(toks) - 7)
    for line in sorted(c):
        print(f"  {line:5d} {c[line]:4d}  {text.split(chr(10))[line - 1].strip()[:150]}")
