"""Function-level similarity check of the product's host sources against the reference (the round-4 review's method):
8-token shingles, comments stripped, every non-clogs reference source.  Prints, per repository file, the share of its
shingles that occur anywhere in the reference and the reference files that reappear most.  Needs /root/reference (this
container only); nothing here is imported by the product or the tests.

    python tools/shingle_overlap.py [paths...]        default: host/*.cpp host/*.h pipeline.py u3d.py oracle/*.c
"""
import re
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
K = 8
TOKEN = re.compile(r"[A-Za-z_][A-Za-z_0-9]*|\d+\.?\d*[fFuUlL]*|\"(?:\\.|[^\"\\])*\"|[^\sA-Za-z_0-9]")


def strip_comments(text: str, python: bool) -> str:
    if python:
        text = re.sub(r'"""(?:.|\n)*?"""', " ", text)
        return re.sub(r"#[^\n]*", " ", text)
    text = re.sub(r"/\*(?:.|\n)*?\*/", " ", text)
    return re.sub(r"//[^\n]*", " ", text)


def shingles(path: Path):
    text = strip_comments(path.read_text(errors="replace"), path.suffix == ".py")
    toks = TOKEN.findall(text)
    return {tuple(toks[i:i + K]) for i in range(len(toks) - K + 1)}


def main():
    args = [Path(a) for a in sys.argv[1:]]
    if not args:
        pkg = next(REPO.glob("correlated*"))
        args = sorted(pkg.glob("host/*.cpp")) + sorted(pkg.glob("host/*.h")) + [pkg / "pipeline.py", pkg / "u3d.py"] + sorted((REPO / "oracle").glob("*.c"))
    ref_files = [p for p in REF.rglob("*") if p.suffix in (".cpp", ".h", ".cl", ".frag", ".hpp") and "ext/clogs" not in str(p)]
    ref = {p: shingles(p) for p in ref_files}
    everything = set().union(*ref.values())
    for a in args:
        s = shingles(a)
        if not s:
            continue
        shared = s & everything
        line = f"{a.relative_to(REPO) if a.is_absolute() else a}: {100 * len(shared) / len(s):5.1f} % of {len(s)} shingles shared"
        tops = sorted(((len(s & r) / max(1, len(r)), p) for p, r in ref.items()), reverse=True)[:3]
        line += "; reference files reappearing: " + ", ".join(f"{p.name} {100 * f:.0f} %" for f, p in tops if f > 0.02)
        print(line)


if __name__ == "__main__":
    main()
