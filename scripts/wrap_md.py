"""Development aid: reflow a markdown file to a column limit (default 120).  Prose paragraphs, list items (hanging indent) and block quotes are re-wrapped;
headings, code fences and tables whose rows fit stay as they are; a table with rows longer than `--table-limit` (default 360) cannot be wrapped as a table and
is turned into a list - one item per row, "**first cell** - second cell - ..." with the header row as a legend line - which then wraps like any other list
(tables of five or more columns are matrices of figures and stay tables whatever their width).
usage: python scripts/wrap_md.py FILE [--width 120] [--table-limit 360] [--in-place]"""
import re
import sys
import textwrap


def split_row(line):
    cells, cur, depth, i = [], "", 0, 0
    body = line.strip()
    body = body[1:] if body.startswith("|") else body
    body = body[:-1] if body.endswith("|") else body
    while i < len(body):
        ch = body[i]
        if ch == "\\" and i + 1 < len(body) and body[i + 1] == "|":
            cur += "|"
            i += 2
            continue
        if ch == "`":
            depth ^= 1
        if ch == "|" and not depth:
            cells.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    cells.append(cur.strip())
    return cells


def wrap_block(text, width, first, rest):
    return textwrap.fill(" ".join(text.split()), width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def reflow(src, width=120, table_limit=360, matrix_columns=5):
    lines = src.split("\n")
    out, i, fence = [], 0, False
    item = re.compile(r"^(\s*)([*+-]|\d+[.)])\s+")
    while i < len(lines):
        ln = lines[i]
        if ln.strip().startswith("```"):
            fence = not fence
            out.append(ln)
            i += 1
            continue
        if fence or not ln.strip() or ln.lstrip().startswith("#") or ln.strip() in ("---", "***"):
            out.append(ln)
            i += 1
            continue
        if ln.lstrip().startswith("|"):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            table = lines[i:j]
            if max(len(r) for r in table) <= table_limit or len(split_row(table[0])) >= matrix_columns:
                out.extend(table)   # (a matrix of figures stays a table whatever its width: as a list it would lose its columns)
            else:
                indent = re.match(r"^\s*", table[0]).group(0)
                rows = [split_row(r) for r in table]
                header = rows[0] if len(rows) > 1 and re.fullmatch(r"[\s|:-]+", table[1]) else None
                body = rows[2:] if header else rows
                if header and any(header):
                    out.append(wrap_block("(" + " · ".join(h for h in header if h) + ")", width, indent, indent))
                    out.append("")
                for r in body:
                    cells = [c for c in r]
                    head = cells[0] if cells and cells[0] else "-"
                    rest = [c for c in cells[1:] if c]
                    text = ("**" + head.strip("*") + "**" if head != "-" else "-") + ("".join(" — " + c for c in rest))
                    out.append(wrap_block(text, width, indent + "* ", indent + "  "))
                out.append("")
            i = j
            continue
        m = item.match(ln)
        first = m.group(0) if m else re.match(r"^\s*(>\s*)?", ln).group(0)
        rest = " " * len(first) if m else first
        para = [ln[len(first):]]
        i += 1
        while i < len(lines):
            nx = lines[i]
            if not nx.strip() or nx.lstrip().startswith(("#", "|", "```")) or item.match(nx):
                break
            if m is None and re.match(r"^\s*", nx).group(0) != re.match(r"^\s*", ln).group(0) and len(nx) - len(nx.lstrip()) >= 4:
                break
            para.append(nx.strip())
            i += 1
        out.append(wrap_block(" ".join(para), width, first, rest))
    return "\n".join(out)


if __name__ == "__main__":
    args = sys.argv[1:]
    path = args[0]
    width = int(args[args.index("--width") + 1]) if "--width" in args else 120
    tl = int(args[args.index("--table-limit") + 1]) if "--table-limit" in args else 360
    text = reflow(open(path).read(), width, tl)
    if "--in-place" in args:
        open(path, "w").write(text)
    else:
        sys.stdout.write(text)
