#!/usr/bin/env python3
"""Re-flows Markdown to lines of at most WIDTH characters without changing what it renders to (much): paragraphs and list
items are re-wrapped with a hanging indent; fenced code is left alone; a table whose rows do not fit becomes a list -- one
bullet per row (first cell in bold), one sub-bullet per further cell, labelled with the column's header.
usage: wrap_md.py IN.md [OUT.md] [--width 160]"""
import re
import sys
import textwrap

WIDTH = 160


def wrap(text, first, rest):
    text = " ".join(text.split())
    if not text:
        return [first.rstrip()]
    # break_on_hyphens off: paths and options stay whole; long tokens (a URL, a code span) are allowed to overflow
    return textwrap.wrap(text, width=WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def split_row(line):
    cells, cur, code = [], "", False
    body = line.strip()
    if body.startswith("|"):
        body = body[1:]
    if body.endswith("|") and not body.endswith("\\|"):
        body = body[:-1]
    i = 0
    while i < len(body):
        ch = body[i]
        if ch == "`":
            code = not code
        if ch == "\\" and i + 1 < len(body) and body[i + 1] == "|":
            cur += "|"
            i += 2
            continue
        if ch == "|" and not code:
            cells.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    cells.append(cur.strip())
    return cells


def table_to_list(rows):
    header = split_row(rows[0])
    out = []
    for r in rows[2:]:
        cells = split_row(r)
        head = cells[0] if cells and cells[0] else "—"
        label0 = header[0].strip()
        out += wrap(("**" + head + "**") if not label0 or label0 in ("#",) else f"**{label0} {head}**" if len(label0) <= 12 and len(head) <= 40 else "**" + head + "**", "- ", "  ")
        for h, c in zip(header[1:], cells[1:]):
            if c:
                out += wrap(f"*{h}*: {c}" if h else c, "  - ", "    ")
    return out


def reflow(lines):
    out, i, n = [], 0, len(lines)
    while i < n:
        line = lines[i].rstrip("\n")
        if re.match(r"^\s*(```|~~~)", line):       # fenced code: verbatim
            fence = line.strip()[:3]
            out.append(line)
            i += 1
            while i < n and not lines[i].strip().startswith(fence):
                out.append(lines[i].rstrip("\n"))
                i += 1
            if i < n:
                out.append(lines[i].rstrip("\n"))
                i += 1
            continue
        if line.lstrip().startswith("|") and i + 1 < n and re.match(r"^\s*\|?\s*:?-{2,}", lines[i + 1]):   # a table
            rows = []
            while i < n and lines[i].lstrip().startswith("|"):
                rows.append(lines[i].rstrip("\n"))
                i += 1
            if max(len(r) for r in rows) <= WIDTH:
                out += rows
            else:
                out += table_to_list(rows)
                out.append("")
            continue
        if not line.strip() or re.match(r"^\s*(#|---+\s*$|===+\s*$|<)", line):
            out.append(line)
            i += 1
            continue
        # a paragraph or a list item: gather its continuation lines
        m = re.match(r"^(\s*)([-*+]|\d+[.)])\s+", line)
        if m:
            first = m.group(0)
            rest = " " * len(first)
            text = line[len(first):]
        else:
            ind = re.match(r"^\s*", line).group(0)
            first = rest = ind
            text = line.strip()
        i += 1
        while i < n:
            nxt = lines[i].rstrip("\n")
            if (not nxt.strip() or re.match(r"^\s*([-*+]|\d+[.)])\s+", nxt) or re.match(r"^\s*(#|```|~~~|\||<)", nxt)):
                break
            text += " " + nxt.strip()
            i += 1
        out += wrap(text, first, rest)
    return out


def main():
    global WIDTH
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--width" in sys.argv:
        WIDTH = int(sys.argv[sys.argv.index("--width") + 1])
        args = [a for a in args if a != str(WIDTH)]
    src = args[0]
    dst = args[1] if len(args) > 1 else src
    with open(src) as f:
        lines = f.readlines()
    res = reflow(lines)
    with open(dst, "w") as f:
        f.write("\n".join(res).rstrip("\n") + "\n")
    over = [k + 1 for k, l in enumerate(res) if len(l) > WIDTH]
    print(f"{dst}: {len(res)} lines, {len(over)} over {WIDTH} characters" + (f" (first: {over[:5]})" if over else ""))


if __name__ == "__main__":
    main()
