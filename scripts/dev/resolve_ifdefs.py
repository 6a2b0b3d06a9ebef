"""Round-5 clean-up tool: resolve preprocessor conditionals whose macros have ONE shipped value (the A/B arms of
earlier rounds) and drop the dead arms.  usage: resolve_ifdefs.py FILE NAME=VALUE ... NAME= (undefined)
A default block `#ifndef X / #define X v / #endif` becomes the single line `#define X v`."""
import re
import sys


def evaluate(cond, known):
    """Value of a preprocessor condition if every identifier in it is known, else None."""
    expr = cond.split("//")[0].strip()
    names = set(re.findall(r"[A-Za-z_]\w*", expr)) - {"defined"}
    if not names or not names <= set(known):
        return None
    def sub_defined(m):
        return "1" if known[m.group(1)] is not None else "0"
    expr = re.sub(r"defined\s*\(?\s*(\w+)\s*\)?", sub_defined, expr)
    for n in names:
        expr = re.sub(r"\b%s\b" % n, str(known[n] if known[n] is not None else 0), expr)
    expr = expr.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return bool(eval(expr))


def resolve(lines, known):
    out, stack = [], []   # stack entries: [kind, emitting_parent, taken, resolved]
    i = 0
    while i < len(lines):
        l = lines[i]
        s = l.strip()
        emit = all(e[1] for e in stack) if stack else True
        live = all((not e[3]) or e[2] for e in stack)   # inside only live arms of resolved blocks
        m = re.match(r"#\s*(ifdef|ifndef|if)\s+(.*)", s)
        if m:
            kind, cond = m.group(1), m.group(2)
            name = cond.split()[0] if kind != "if" else None
            val = None
            if kind == "ifdef" and name in known:
                val = known[name] is not None
            elif kind == "ifndef" and name in known:
                val = known[name] is None
                # default block: #ifndef X / #define X v / #endif  -> keep the define alone
                if known[name] is not None and i + 2 < len(lines) and re.match(r"#\s*define\s+%s\b" % name, lines[i + 1].strip()):
                    j = i + 2
                    while j < len(lines) and not lines[j].strip().startswith("#"):
                        j += 1
                    if lines[j].strip().startswith("#endif") and live:
                        out.extend(lines[i + 1:j])
                        i = j + 1
                        continue
            elif kind == "if":
                val = evaluate(cond, known)
            if val is None:
                stack.append([kind, True, True, False])
                if live:
                    out.append(l)
            else:
                stack.append([kind, True, val, True])
            i += 1
            continue
        if re.match(r"#\s*else\b", s):
            e = stack[-1]
            if e[3]:
                e[2] = not e[2]
            elif all((not x[3]) or x[2] for x in stack[:-1]):
                out.append(l)
            i += 1
            continue
        if re.match(r"#\s*endif\b", s):
            e = stack.pop()
            if not e[3] and all((not x[3]) or x[2] for x in stack):
                out.append(l)
            i += 1
            continue
        if live:
            out.append(l)
        i += 1
    return out


if __name__ == "__main__":
    path = sys.argv[1]
    known = {}
    for a in sys.argv[2:]:
        k, v = a.split("=", 1)
        known[k] = None if v == "" else int(v)
    src = open(path).read().split("\n")
    open(path, "w").write("\n".join(resolve(src, known)))
