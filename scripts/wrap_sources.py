#!/usr/bin/env python3
"""Wrap over-long lines of C / C++ / HIP sources at a column limit without touching a token (kernel sources at
<= 160 columns, no behaviour change).  Whitespace between tokens and the position of comments are all that changes:

  * a line that is code followed by a trailing /* comment */ : the comment moves to its own line above, at the code's indent;
  * a comment line (/* ..., * ..., // ...): reflowed at word boundaries, the continuation lines carrying the comment's own prefix;
  * a code line: broken after the last `; `, `, `, ` && `, ` || `, ` ? `, ` : `, ` + `, ` = ` ... before the limit that is outside string / char
    literals and comments, the continuation indented by 4 more; inside a macro definition the break carries a trailing backslash.

    python scripts/wrap_sources.py [--limit 160] [--check] file ...
--check prints the lines that would still be too long (none expected) and changes nothing."""
import argparse
import re
import sys

BREAKS = ["; ", ", ", " && ", " || ", " ? ", " : ", " + ", " - ", " * ", " = ", " << ", " | ", " < ", " > ", " == ", ") "]


def split_trailing_comment(line):
    """(code, comment) when the line is code followed by ONE trailing /* ... */ that closes on the line; else (line, None)."""
    s = line.rstrip("\n")
    if not s.rstrip().endswith("*/"):
        return s, None
    i = s.rfind("/*")
    if i <= 0 or "*/" in s[i + 2:-2]:
        return s, None
    code = s[:i].rstrip()
    if not code.strip() or code.rstrip().endswith("\\"):
        return s, None
    # the comment opener must not sit inside a string literal
    if code.count('"') % 2 == 1:
        return s, None
    return code, s[i:].strip()


def outside_literals(s):
    """mask[i] = True where position i is outside string / char literals and comments."""
    mask = [True] * len(s)
    i, n = 0, len(s)
    while i < n:
        c = s[i]
        if c in "\"'":
            j = i + 1
            while j < n and s[j] != c:
                j += 2 if s[j] == "\\" else 1
            for k in range(i, min(j + 1, n)):
                mask[k] = False
            i = j + 1
        elif s.startswith("/*", i):
            j = s.find("*/", i + 2)
            j = n if j < 0 else j + 2
            for k in range(i, j):
                mask[k] = False
            i = j
        elif s.startswith("//", i):
            for k in range(i, n):
                mask[k] = False
            break
        else:
            i += 1
    return mask


def wrap_comment(s, limit):
    m = re.match(r"^(\s*)(/\*+|\*|//+)(\s*)(.*)$", s)
    if not m:
        return [s]
    indent, opener, gap, text = m.groups()
    closes = opener.startswith("/*") or opener == "*"
    first = indent + opener + (gap or " ")
    cont = indent + (" * " if opener.startswith("/*") or opener == "*" else opener + " ")
    if opener.startswith("/*") and not text.rstrip().endswith("*/"):
        cont = indent + " * "
    words, out, cur = text.split(" "), [], first
    for w in words:
        if len(cur) + len(w) > limit and cur.strip() not in ("/*", "*", "//") and len(cur) > len(cont):
            out.append(cur.rstrip())
            cur = cont
        cur += w + " "
    out.append(cur.rstrip())
    return out if closes or opener.startswith("//") else [s]


def wrap_code(s, limit, in_macro=False):
    out = []
    # a break inside a macro definition needs a trailing backslash: the line continues one (it ends with a backslash), starts one
    # (#define ... on a single line) or is the last line of one (the line before it ended with a backslash)
    macro = s.rstrip().endswith("\\") or s.lstrip().startswith("#define") or in_macro
    indent = len(s) - len(s.lstrip())
    cont_indent = " " * (indent + 4)
    guard = 0
    while len(s) > limit and guard < 20:
        guard += 1
        mask = outside_literals(s)
        room = limit - (2 if macro else 0)
        best = -1
        for b in BREAKS:
            start = 0
            while True:
                i = s.find(b, start)
                if i < 0 or i + len(b) > room:
                    break
                if mask[i] and mask[i + len(b) - 1] and i > indent + 8:
                    # never split `for (a; b; c)` heads at their semicolons, nor template angle brackets at " < " / " > "
                    if not (b == "; " and s[:i].count("(") > s[:i].count(")")):
                        best = max(best, i + len(b))
                start = i + 1
            if best >= 0 and b in ("; ", ", "):
                break
        if best < 0:
            break
        head, tail = s[:best].rstrip(), s[best:].lstrip()
        if not tail or tail == "\\":
            break
        out.append(head + (" \\" if macro and not head.endswith("\\") else ""))
        s = cont_indent + tail
    out.append(s)
    return out


def process(text, limit):
    out = []
    prev_continues = False
    for line in text.split("\n"):
        in_macro, prev_continues = prev_continues, line.rstrip().endswith("\\")
        if len(line) <= limit:
            out.append(line)
            continue
        stripped = line.lstrip()
        if stripped.startswith(("/*", "* ", "//")) and not (stripped.startswith("/*") and "*/" in stripped[2:-2] and not stripped.endswith("*/")):
            out += wrap_comment(line, limit)
            continue
        code, comment = split_trailing_comment(line)
        if comment is not None:
            ind = " " * (len(code) - len(code.lstrip()))
            out += wrap_comment(ind + comment, limit)
            out += wrap_code(code, limit, in_macro) if len(code) > limit else [code]
            continue
        if stripped.startswith("#") and not stripped.startswith("#define"):
            out.append(line)
            continue
        out += wrap_code(line, limit, in_macro)
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--limit", type=int, default=160)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("files", nargs="+")
    a = ap.parse_args()
    left = 0
    for f in a.files:
        text = open(f).read()
        new = process(text, a.limit)
        for k, line in enumerate(new.split("\n")):
            if len(line) > a.limit:
                left += 1
                if a.check:
                    print("%s:%d: %d columns" % (f, k + 1, len(line)))
        if not a.check and new != text:
            open(f, "w").write(new)
    print("lines still over the limit: %d" % left)


if __name__ == "__main__":
    main()
