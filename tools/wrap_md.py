#!/usr/bin/env python
"""Re-flows a markdown file to 146 characters (<= 160 bytes per line with the usual share of multi-byte signs): paragraphs and list items are joined and wrapped with a hanging indent, tables whose rows would be longer than the limit
become bullet lines ("* cell — cell — cell"), headings / fenced code / short tables are left alone.  usage: python tools/wrap_md.py FILE [...]"""
import re
import sys
import textwrap

LIMIT = 146
ITEM = re.compile(r"^(\s*)([*-]|\d+\.)\s+")


def wrap(text, first="", rest=""):
    return textwrap.wrap(text, LIMIT, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def convert(lines):
    out, i, n = [], 0, len(lines)
    while i < n:
        line = lines[i]
        if line.startswith("```"):
            j = i + 1
            while j < n and not lines[j].startswith("```"):
                j += 1
            out.extend(lines[i:j + 1]); i = j + 1; continue
        if not line.strip() or line.startswith("#"):
            out.append(line); i += 1; continue
        if line.startswith("|"):
            j = i
            while j < n and lines[j].startswith("|"):
                j += 1
            block = lines[i:j]
            if max(len(b) for b in block) <= LIMIT:
                out.extend(block)
            else:
                for b in block:
                    cells = [c.strip() for c in b.strip().strip("|").split("|")]
                    if all(re.fullmatch(r":?-+:?", c) for c in cells):
                        continue
                    out.extend(wrap(" — ".join(c for c in cells if c), "* ", "  "))
            i = j; continue
        m = ITEM.match(line)
        if m:      # a list item and its continuation lines (indented deeper than the bullet, or plain text right below)
            first = m.group(0)
            text = [line[len(first):].strip()]
            j = i + 1
            while j < n and lines[j].strip() and not ITEM.match(lines[j]) and not lines[j].startswith(("#", "|", "```")):
                text.append(lines[j].strip()); j += 1
            out.extend(wrap(" ".join(text), first, " " * len(first)))
            i = j; continue
        text = [line.strip()]
        ind = re.match(r"^\s*", line).group(0)
        j = i + 1
        while j < n and lines[j].strip() and not ITEM.match(lines[j]) and not lines[j].startswith(("#", "|", "```")):
            text.append(lines[j].strip()); j += 1
        out.extend(wrap(" ".join(text), ind, ind))
        i = j
    return out


for path in sys.argv[1:]:
    src = open(path).read().split("\n")
    res = convert(src)
    open(path, "w").write("\n".join(res))
    print(path, len(res), "lines, longest", max(len(l) for l in res))
