#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file to a column limit (default 118) without touching tables, code fences or headings.
List items and their continuation lines keep their indentation.      python tools/reflow_md.py FILE [width]"""
import re
import sys
import textwrap


def reflow(text: str, width: int = 118) -> str:
    out, para, indent_first, indent_rest = [], [], "", ""
    in_code = False

    def flush():
        nonlocal para
        if para:
            body = " ".join(s.strip() for s in para)
            out.extend(textwrap.wrap(body, width=width, initial_indent=indent_first, subsequent_indent=indent_rest,
                                     break_long_words=False, break_on_hyphens=False) or [""])
            para = []

    for line in text.split("\n"):
        if line.strip().startswith("```"):
            flush(); in_code = not in_code; out.append(line); continue
        is_table = line.lstrip().startswith("|") and line.rstrip().endswith("|") and line.count("|") >= 3
        if in_code or is_table or line.startswith("#") or not line.strip():
            flush(); out.append(line); continue
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", line)
        if m:                                                   # a new list item
            flush()
            indent_first = m.group(0)
            indent_rest = " " * len(m.group(0))
            para = [line[len(m.group(0)):]]
            continue
        lead = len(line) - len(line.lstrip())
        if not para:
            indent_first = indent_rest = " " * lead
        para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    src = open(path).read()
    open(path, "w").write(reflow(src, width))
