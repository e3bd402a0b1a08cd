#!/usr/bin/env python3
"""Which arrays a kernel of blom_amd/csrc reads and writes, from its source text: the fields V.f[F_*], the integer fields V.m[I_*], the
work-space slots WK(V, slot) / WK2(V, slot) (slot names resolved to numbers through the file's enums and #defines, so that slots of
different stages compare).  An array counts as WRITTEN when it is bound to a mutable pointer (gd_t / gi_t) or stored through directly;
every other mention is a read.  Used by tests/test_side_by_side_sets.py to check the pairs blomgpu_step runs side by side on its two
streams, and from the command line:  tools/kernel_rw_sets.py blom_amd/csrc/stage_x.hip k_a k_b ..."""
import re, sys

TOK = re.compile(r"\b(F_\w+|I_\w+)\b|\bWK\(\s*\w+\s*,([^()]*)\)|\bWK2\(\s*\w+\s*,([^()]*)\)")


def constants(src):
    """names of enum members and integer #defines -> value"""
    val = {}
    for m in re.finditer(r"#define\s+(\w+)\s+(\d+)\s*$", src, re.M):
        val[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"enum\s*\w*\s*\{([^}]*)\}", src):
        nxt = 0
        for item in m.group(1).split(","):
            item = re.sub(r"//.*|/\*.*?\*/", "", item).strip()
            if not item:
                continue
            if "=" in item:
                name, e = [x.strip() for x in item.split("=", 1)]
                nxt = int(e) if e.isdigit() else val.get(e, nxt)
            else:
                name = item
            val[name] = nxt
            nxt += 1
    return val


def bodies(src):
    out = {}
    for m in re.finditer(r"__global__[^;{]*?\bvoid\s+(\w+)\s*\(", src):
        i = src.index("{", m.end())
        d, j = 0, i
        while True:
            d += src[j] == "{"
            d -= src[j] == "}"
            if d == 0:
                break
            j += 1
        out[m.group(1)] = src[i:j + 1]
    return out


def _names(text, val):
    for m in TOK.finditer(text):
        if m.group(1):
            yield m.group(1)
        else:                                            # the slot expression may choose between slots (isv ? W_DV2 : W_DU2): every name in it
            pre, expr = ("wk", m.group(2)) if m.group(2) is not None else ("wk2", m.group(3))
            ids = re.findall(r"[A-Za-z_]\w*", expr)
            known = [i for i in ids if i in val]
            for i in known or ids or [expr.strip()]:
                yield "%s:%s" % (pre, val.get(i, i))


# device helpers that store through a pointer argument (blomgpu_internal.h): name -> positions of those arguments
OUT_ARGS = {"column_scan": [2]}


def _call_args(text, start):
    """the arguments of the call whose opening parenthesis is at text[start]"""
    args, d, cur = [], 0, ""
    for ch in text[start:]:
        if ch == "(":
            d += 1
            if d == 1:
                continue
        elif ch == ")":
            d -= 1
            if d == 0:
                args.append(cur)
                return args
        if ch == "," and d == 1:
            args.append(cur)
            cur = ""
        else:
            cur += ch
    return args


def rw_sets(path, kernel):
    src = open(path).read()
    val = constants(src)
    body = re.sub(r"//[^\n]*", "", bodies(src)[kernel])
    reads, writes = set(_names(body, val)), set()
    for stmt in body.split(";"):
        m = re.search(r"\b(gd_t|gi_t|double\s*\*|int\s*\*)(?:\s*const)?\s*(?:__restrict__\s*)?\w+\s*=(.*)", stmt, re.S)
        if m:                                            # mutable pointer(s): every array named on the right-hand side(s)
            writes |= set(_names(m.group(2), val))
        for m in re.finditer(r"(?:V\.[fm]\[[^\]]*\]|WK2?\([^)]*\))(?:\s*\+[^;=\[]*)?\[[^;]*?\]\s*(?:=(?!=)|[-+*/]=)", stmt):
            writes |= set(_names(m.group(0).split("]")[0] + "]" if m.group(0).startswith("V.") else m.group(0), val))
    for fn, outs in OUT_ARGS.items():
        for m in re.finditer(r"\b%s\s*\(" % fn, body):
            args = _call_args(body, m.end() - 1)
            for o in outs:
                if o < len(args):
                    writes |= set(_names(args[o], val))
    return reads - writes, writes


if __name__ == "__main__":
    for k in sys.argv[2:]:
        r, w = rw_sets(sys.argv[1], k)
        print(k, "\n  reads ", " ".join(sorted(r)), "\n  writes", " ".join(sorted(w)))
