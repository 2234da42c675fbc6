"""Resolve a fixed set of preprocessor macros in a source file (a small `unifdef`): conditionals whose condition only names
macros of the given set are evaluated and removed (the live branch stays), `#ifndef X / #define X v / #endif` default blocks of
those macros disappear, everything else passes through untouched.  Macro uses inside C++ expressions are NOT rewritten (grep for
them afterwards).
    python tools/unifdef_lite.py file.hip NAME=1 OTHER=0 UNDEFINED_ONE= > out
Round 6 used it to take the measured-and-rejected experiment branches out of csrc/decoder.hip; the object code before and after
is identical (the ISA of both parts was diffed)."""
import re
import sys


def evaluate(cond, defs):
    """-> True / False, or None when the condition names a macro outside `defs`."""
    names = set(re.findall(r"[A-Za-z_]\w*", re.sub(r"defined\s*\(\s*\w+\s*\)|defined\s+\w+", "", cond))) - {"defined"}
    dnames = set(re.findall(r"defined\s*\(?\s*(\w+)", cond))
    if not (names | dnames) <= set(defs):
        return None
    expr = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", lambda m: "1" if defs[m.group(1) or m.group(2)] is not None else "0", cond)
    expr = re.sub(r"[A-Za-z_]\w*", lambda m: str(defs[m.group(0)] if defs[m.group(0)] not in (None, "") else 0), expr)
    expr = expr.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return bool(eval(expr))


def main(path, defs):
    lines = open(path).read().split("\n")
    out = []
    # stack entries: [resolved?, currently_live, any_branch_taken, parent_live]
    stack = []
    i = 0
    live = lambda: all(s[1] for s in stack if s[0]) if stack else True
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", ln)
        if not m:
            if live():
                out.append(ln)
            i += 1
            continue
        kind, rest = m.group(1), re.sub(r"//.*", "", m.group(2)).strip()
        if kind in ("ifdef", "ifndef", "if"):
            if kind == "ifdef":
                val = (defs[rest] is not None) if rest in defs else None
            elif kind == "ifndef":
                val = (defs[rest] is None) if rest in defs else None
                # default block of a resolved macro:  #ifndef X / #define X v [comment] / #endif  -> gone
                if rest in defs and i + 2 < len(lines) and re.match(r"\s*#\s*define\s+%s\b" % re.escape(rest), lines[i + 1]) and \
                        re.match(r"\s*#\s*endif", lines[i + 2]):
                    i += 3
                    continue
            else:
                val = evaluate(rest, defs)
            if val is None:
                stack.append([False, True, True])
                if live():
                    out.append(ln)
            else:
                stack.append([True, val, val])
        elif kind == "elif":
            top = stack[-1]
            if not top[0]:
                if live():
                    out.append(ln)
            else:
                val = evaluate(rest, defs)
                if val is None:
                    raise SystemExit(f"{path}:{i + 1}: #elif with unknown macros after a resolved #if")
                top[1] = (not top[2]) and val
                top[2] = top[2] or val
        elif kind == "else":
            top = stack[-1]
            if not top[0]:
                if live():
                    out.append(ln)
            else:
                top[1] = not top[2]
                top[2] = True
        else:
            top = stack.pop()
            if not top[0] and live():
                out.append(ln)
        i += 1
    sys.stdout.write("\n".join(out))


if __name__ == "__main__":
    d = {}
    for a in sys.argv[2:]:
        k, _, v = a.partition("=")
        d[k] = None if (v == "" and "=" in a) else v
    main(sys.argv[1], d)
