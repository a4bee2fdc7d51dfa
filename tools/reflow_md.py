"""re-wrap the paragraphs and bullets of a markdown file to <= 118 characters (tables, headings, code fences untouched): python tools/reflow_md.py FILE"""
import re, sys, textwrap
fn = sys.argv[1]
src = open(fn).read().split("\n")
out, para, fence = [], [], False
def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*(?:[*-]|\d+\.)\s+)", first)
    ind = m.group(1) if m else ""
    body = " ".join([first[len(ind):]] + [l.strip() for l in para[1:]])
    out.extend(textwrap.wrap(body, width=118, initial_indent=ind, subsequent_indent=" " * len(ind), break_long_words=False, break_on_hyphens=False))
    para = []
for l in src:
    if l.startswith("```"):
        flush(); fence = not fence; out.append(l)
    elif fence or l.startswith("|") or l.startswith("#") or l.strip() == "":
        flush(); out.append(l)
    elif re.match(r"^\s*(?:[*-]|\d+\.)\s+", l):
        flush(); para = [l]
    else:
        para.append(l)
flush()
open(fn, "w").write("\n".join(out))
