#!/usr/bin/env python3
"""strip_switches.py FILE... -D NAME=VALUE ... -U NAME ...   -- a minimal `unifdef` (none in the image).

Resolves the preprocessor conditionals of FILE whose expression uses only the given names (-D: defined with an integer
value, -U: undefined) and rewrites FILE in place: the taken branch stays, the others and the directive lines go; the
`#ifndef NAME / #define NAME v / #endif` default blocks of -D names are removed, and so are stray `#define NAME` /
`#undef NAME` lines of them.  Conditionals on anything else are left untouched.  Uses of a name OUTSIDE directives are
reported, not rewritten (round 6: the experiment switches of the product kernels)."""
import re
import sys


def parse(argv):
    files, vals = [], {}
    it = iter(argv)
    for a in it:
        if a == "-D":
            k, _, v = next(it).partition("=")
            vals[k] = int(v or "1", 0)
        elif a == "-U":
            vals[next(it)] = None
        else:
            files.append(a)
    return files, vals


TOK = re.compile(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)|([A-Za-z_]\w*)")


def evaluate(expr, vals):
    """-> True / False, or None when the expression uses a name we do not know"""
    expr = re.sub(r"//.*$", "", expr).strip()
    unknown = False

    def sub(m):
        nonlocal unknown
        d = m.group(1) or m.group(2)
        if d:
            if d not in vals:
                unknown = True
                return "0"
            return "1" if vals[d] is not None else "0"
        n = m.group(3)
        if n not in vals:
            unknown = True
            return "0"
        return str(vals[n] if vals[n] is not None else 0)

    py = TOK.sub(sub, expr)
    if unknown:
        return None
    py = py.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return bool(eval(py, {"__builtins__": {}}))


def strip(text, vals):
    out, stack = [], []  # stack of dicts: known (bool), taken (bool so far), active (emit?), parent_active
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        # continuation lines of a directive belong to it
        full, j = ln, i
        while full.rstrip().endswith("\\") and j + 1 < len(lines):
            j += 1
            full += "\n" + lines[j]
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif|define|undef)\b(.*)", ln, re.S)
        active = all(f["emit"] for f in stack)
        if not m:
            if active:
                out.append(ln)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            if kind == "if":
                v = evaluate(rest, vals)
            else:
                name = rest.split()[0]
                v = None if name not in vals else ((vals[name] is not None) == (kind == "ifdef"))
            # the default block  #ifndef NAME / #define NAME v / #endif  of a known, defined name: drop it whole
            if kind == "ifndef" and v is False:
                pass
            stack.append({"known": v is not None, "emit": True if v is None else v, "taken": bool(v), "kept": v is None})
            if v is None and active:
                out.append(ln)
        elif kind == "elif":
            f = stack[-1]
            if f["known"]:
                if f["taken"]:
                    f["emit"] = False
                else:
                    v = evaluate(rest, vals)
                    if v is None:
                        raise SystemExit(f"line {i + 1}: #elif on unknown names after a resolved #if: not supported")
                    f["emit"], f["taken"] = v, v
            elif all(g["emit"] for g in stack[:-1]):
                out.append(ln)
        elif kind == "else":
            f = stack[-1]
            if f["known"]:
                f["emit"] = not f["taken"]
                f["taken"] = True
            elif all(g["emit"] for g in stack[:-1]):
                out.append(ln)
        elif kind == "endif":
            f = stack.pop()
            if not f["known"] and all(g["emit"] for g in stack):
                out.append(ln)
        else:  # define / undef
            name = rest.split()[0].split("(")[0]
            if active and not (name in vals):
                out.extend(full.split("\n"))
            i = j + 1
            continue
        i += 1
    if stack:
        raise SystemExit("unbalanced conditionals")
    return "\n".join(out)


def main():
    files, vals = parse(sys.argv[1:])
    for path in files:
        src = open(path).read()
        new = strip(src, vals)
        # collapse the blank-line runs the removed blocks leave behind
        new = re.sub(r"\n{3,}", "\n\n", new)
        open(path, "w").write(new)
        for k in vals:
            for n, ln in enumerate(new.split("\n"), 1):
                if re.search(rf"\b{k}\b", re.sub(r"//.*$", "", ln)):
                    print(f"{path}:{n}: still uses {k}: {ln.strip()[:110]}")


if __name__ == "__main__":
    main()
