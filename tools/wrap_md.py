#!/usr/bin/env python3
"""Hard-wraps the prose of markdown files at a column (default 130): paragraphs and list items are re-flowed, tables, headings,
code fences, indented code and lines that carry a trailing double space are left alone.   tools/wrap_md.py [--width N] FILE..."""
import re, sys, textwrap

BULLET = re.compile(r"^(\s*)([*\-]|\d+\.|\([ivxlc]+\)|\([a-z0-9]\))\s+")


def flow(lines, width):
    """lines: one paragraph or one list item (first line may start with a bullet); returns wrapped lines"""
    first = lines[0]
    m = BULLET.match(first)
    if m:
        head = first[:m.end()]
        # ("(i) ..." enumerations inside a paragraph start a line of their own and flow without a hanging indent)
        indent = " " * len(m.group(1)) + ("" if m.group(2).startswith("(") else "  ")
        # (continuation lines of an item keep the indentation the file uses for them, when it is consistent)
        conts = [len(l) - len(l.lstrip()) for l in lines[1:]]
        if conts and min(conts) == max(conts) and conts[0] > len(m.group(1)):
            indent = " " * conts[0]
        text = " ".join([first[m.end():].strip()] + [l.strip() for l in lines[1:]])
        return textwrap.wrap(text, width=width, initial_indent=head, subsequent_indent=indent, break_long_words=False,
                             break_on_hyphens=False)
    lead = first[:len(first) - len(first.lstrip())]
    text = " ".join(l.strip() for l in lines)
    return textwrap.wrap(text, width=width, initial_indent=lead, subsequent_indent=lead, break_long_words=False, break_on_hyphens=False)


def wrap_file(path, width):
    src = open(path).read().split("\n")
    out, para, fence = [], [], False

    def flush():
        nonlocal para
        if para:
            if any(len(l) > width for l in para):
                out.extend(flow(para, width))
            else:
                out.extend(para)                           # (a paragraph that already fits is left exactly as it is)
            para = []

    for line in src:
        stripped = line.strip()
        if stripped.startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence or not stripped or stripped.startswith("|") or stripped.startswith("#") or line.startswith("    ") and not para \
                or stripped.startswith("<") or stripped.startswith("{\"") or line.endswith("  "):
            flush(); out.append(line); continue
        if BULLET.match(line) and para:
            flush()
        para.append(line)
    flush()
    new = "\n".join(out)
    if new != "\n".join(src):
        open(path, "w").write(new)
    return sum(1 for l in out if len(l) > width and not l.lstrip().startswith("|"))


if __name__ == "__main__":
    args = sys.argv[1:]
    width = 130
    if args[:1] == ["--width"]:
        width = int(args[1]); args = args[2:]
    for p in args:
        left = wrap_file(p, width)
        print("%s: %d prose lines still over %d columns" % (p, left, width))
