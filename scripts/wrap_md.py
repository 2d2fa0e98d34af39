#!/usr/bin/env python3
"""Wrap the prose of a Markdown file at 120 columns (round-5 review: DESIGN.md had 1 000-character lines).  Tables, fenced
code, headings and indented code are left alone; list items get a hanging indent.  Usage: wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap


def wrap(text, width=120):
    out, para, in_code = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", first)
        if m:
            lead, hang = m.group(0), " " * len(m.group(0))
            body = " ".join([first[len(lead):]] + [p.strip() for p in para[1:]])
            out.extend(textwrap.wrap(body, width=width, initial_indent=lead, subsequent_indent=hang, break_long_words=False,
                                     break_on_hyphens=False) or [lead.rstrip()])
        else:
            ind = re.match(r"^\s*", first).group(0)
            body = " ".join(p.strip() for p in para)
            out.extend(textwrap.wrap(body, width=width, initial_indent=ind, subsequent_indent=ind, break_long_words=False,
                                     break_on_hyphens=False))
        para.clear()

    for line in text.split("\n"):
        if line.strip().startswith("```"):
            flush()
            in_code = not in_code
            out.append(line)
            continue
        if in_code or line.startswith("    ") and not para or line.lstrip().startswith("|") or line.startswith("#") or \
                re.match(r"^\s*([-=*_]{3,})\s*$", line) or line.startswith("{"):
            flush()
            out.append(line)
            continue
        if not line.strip():
            flush()
            out.append("")
            continue
        if re.match(r"^\s*([-*+]|\d+\.)\s+", line):  # a new list item ends the paragraph before it
            flush()
        para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    src = open(path).read()
    open(path, "w").write(wrap(src, width))
