#!/usr/bin/env python3
"""Make a Markdown file readable at a fixed width: over-long prose lines are wrapped (a list item keeps its hanging
indent; headings and code blocks are left alone) and tables whose cells hold paragraphs are turned into lists
("- **first cell** -- second cell | third cell ...", wrapped).   python tools/wrap_md.py DESIGN.md [width]"""
import re
import sys
import textwrap

path, width = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 118


def wrap(line, out):
    m = re.match(r"^(\s*)((?:[*+-]|\d+\.)\s+)?", line)
    indent, marker = m.group(1), m.group(2) or ""
    body = line[len(indent) + len(marker):]
    for k, w in enumerate(textwrap.wrap(body, width=width - len(indent) - len(marker), break_long_words=False, break_on_hyphens=False)):
        out.append(indent + (marker if k == 0 else " "*len(marker)) + w)


def cells(row):
    return [c.strip() for c in re.split(r"(?<!\\)\|", row.strip().strip("|"))]


lines = open(path).read().split("\n")
out, fence, i = [], False, 0
while i < len(lines):
    line = lines[i]
    if line.lstrip().startswith("```"):
        fence = not fence
    if not fence and line.lstrip().startswith("|"):
        j = i
        while j < len(lines) and lines[j].lstrip().startswith("|"):
            j += 1
        block = lines[i:j]
        if max(len(b) for b in block) > 250:
            head = cells(block[0])
            rows = [cells(b) for b in block[1:] if not re.match(r"^\s*\|[\s:|-]+\|\s*$", b)]
            if any(h for h in head):
                rows = rows if not any(h for h in head) else rows
                if any(h for h in head):
                    wrap("*(" + " / ".join(h for h in head if h) + ")*", out)
                    out.append("")
            for r in rows:
                first = r[0] if r and r[0] else ""
                rest = " | ".join(c for c in r[1:] if c)
                wrap("- " + (f"**{first.strip('*')}** -- " if first else "") + rest, out)
            out.append("")
        else:
            out += block
        i = j
        continue
    if fence or line.lstrip().startswith(("#", "```")) or not line.strip():
        out.append(line)
        i += 1
        continue
    # a paragraph (or one list item with its continuation lines): reflowed as a whole when any of its lines is too long
    is_item = lambda l: re.match(r"^\s*(?:[*+-]|\d+\.)\s+", l) is not None
    j = i + 1
    while j < len(lines) and lines[j].strip() and not lines[j].lstrip().startswith(("|", "#", "```")) and not is_item(lines[j]):
        j += 1
    para = lines[i:j]
    if max(len(l) for l in para) <= width:
        out += para
    else:
        m = re.match(r"^(\s*)((?:[*+-]|\d+\.)\s+)?", para[0])
        wrap(m.group(1) + (m.group(2) or "") + " ".join(l.strip() if k else l[len(m.group(0)):].strip() for k, l in enumerate(para)), out)
    i = j
    continue
open(path, "w").write("\n".join(out))
