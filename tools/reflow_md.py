"""Reflows a markdown file for reading in a terminal / a diff: tables that hold essay-length cells become nested lists (one item per row, one
sub-item per column), and paragraphs / list items are wrapped at WIDTH columns.  Code fences, short tables and headings are left alone.
    python tools/reflow_md.py DESIGN.md [--check]"""
import re
import sys
import textwrap

WIDTH = 150
CELL_MAX = 260      # a table with a cell longer than this becomes a list


def split_row(line):
    cells = re.split(r"(?<!\\)\|", line.strip())
    return [c.strip() for c in cells[1:-1]]


def wrap(text, first, rest):
    return textwrap.fill(text, WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def table_to_list(rows):
    head = split_row(rows[0])
    out = []
    for r in rows[2:]:
        cells = split_row(r)
        if not any(cells):
            continue
        title = cells[0] if cells else ""
        label = "item %s" % title if head[0] == "#" else ("%s — %s" % (head[0], title) if head[0] else title)
        out.append(wrap("**%s**" % label, "* ", "  "))
        for h, c in zip(head[1:], cells[1:]):
            if c:
                out.append(wrap("*%s:* %s" % (h, c) if h else c, "  - ", "    "))
    out.append("")
    return out


def reflow(text):
    lines = text.split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        ln = lines[i]
        if ln.lstrip().startswith("```"):
            fence = not fence
            out.append(ln); i += 1; continue
        if fence or not ln.strip() or ln.startswith("#"):
            out.append(ln); i += 1; continue
        if ln.lstrip().startswith("|"):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            rows = lines[i:j]
            is_table = len(rows) >= 2 and re.match(r"^\s*\|[\s:|-]+\|\s*$", rows[1])
            if is_table and any(len(c) > CELL_MAX for r in rows[2:] for c in split_row(r)):
                out.extend(table_to_list(rows))
            else:
                out.extend(rows)
            i = j; continue
        # a paragraph or a list item: gather its continuation lines (same block, no blank line, no new item / table / heading)
        m = re.match(r"^(\s*)([*+-]|\d+\.)\s+", ln)
        indent = m.group(1) if m else re.match(r"^(\s*)", ln).group(1)
        marker = ln[:m.end()] if m else indent
        body = [ln[len(marker):]]
        j = i + 1
        while j < len(lines):
            nx = lines[j]
            if not nx.strip() or nx.lstrip().startswith(("|", "#", "```")) or re.match(r"^\s*([*+-]|\d+\.)\s+", nx):
                break
            body.append(nx.strip()); j += 1
        text_ = " ".join(b.strip() for b in body)
        rest = " " * len(marker) if m else indent
        out.append(wrap(text_, marker, rest))
        i = j
    return "\n".join(out)


if __name__ == "__main__":
    p = sys.argv[1]
    src = open(p).read()
    dst = reflow(src)
    if "--check" in sys.argv:
        long_ = [len(l) for l in dst.split("\n") if len(l) > WIDTH + 40]
        print("lines", len(src.split("\n")), "->", len(dst.split("\n")), "; lines wider than", WIDTH + 40, ":", len(long_))
    else:
        open(p, "w").write(dst)
