"""One-off source transformation used to introduce csrc/eu_real.h: wraps every floating literal outside comments, strings and
#include lines in R(...).  Kept for the record (and for new code: running it twice is harmless)."""
import re
import sys

LIT = re.compile(r'(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.])')
TOK = re.compile(r'("(?:[^"\\]|\\.)*"|/\*|\*/|//)')


def wrap(text):
    out, in_block = [], False
    for line in text.split("\n"):
        if line.lstrip().startswith("#include"):
            out.append(line)
            continue
        res, pos = [], 0
        # walk the line, tracking block comments across lines
        while pos < len(line):
            if in_block:
                end = line.find("*/", pos)
                if end < 0:
                    res.append(line[pos:]); pos = len(line)
                else:
                    res.append(line[pos:end + 2]); pos = end + 2; in_block = False
                continue
            m = TOK.search(line, pos)
            seg_end = m.start() if m else len(line)
            seg = line[pos:seg_end]

            def sub(mm, seg=seg):
                return mm.group(0) if seg[max(0, mm.start() - 2):mm.start()] == "R(" else "R(%s)" % mm.group(1)
            res.append(LIT.sub(sub, seg))
            if not m:
                break
            tok = m.group(0)
            if tok == "//":
                res.append(line[m.start():]); pos = len(line)
            elif tok == "/*":
                in_block = True; res.append("/*"); pos = m.end()
            elif tok == "*/":
                res.append("*/"); pos = m.end()
            else:
                res.append(tok); pos = m.end()
        out.append("".join(res))
    return "\n".join(out)


for f in sys.argv[1:]:
    s = open(f).read()
    t = wrap(s)
    open(f, "w").write(t)
    print(f, "literals wrapped:", t.count("R(") - s.count("R("))
