"""Hash of the kernel sources as CODE: comments, blank lines and leading / trailing whitespace do not count, so that a commit
which edits a header comment does not invalidate the counter profiles taken from the same kernels (round-5 review, weak 4).
Used by bench.py (is the committed profile from this tree's kernels?) and tools/parse_profiles.py (which stores it).
    python3 tools/csrc_hash.py        -> prints the hash of safe_control_amd/csrc/*.h*"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(text):
    """C++ source without // and /* */ comments (string and character literals are kept as they are)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            j = n if j < 0 else j
            # a line comment continued with a trailing backslash
            while j < n and text[j - 1] == "\\":
                k = text.find("\n", j + 1)
                j = n if k < 0 else k
            i = j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return "".join(out)


def code_lines(text):
    return [ln for ln in (" ".join(l.split()) for l in strip_comments(text).splitlines()) if ln]


def csrc_sha16(root=ROOT):
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(root, "safe_control_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(fn).encode() + b"\0")
        h.update("\n".join(code_lines(open(fn, encoding="utf-8", errors="replace").read())).encode())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha16())
