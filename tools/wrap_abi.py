#!/usr/bin/env python3
"""One-off source transformation (kept for the record and for new entry points): wrap the body of every `extern "C" int`
function of zk-mpc_amd/csrc/*.hip in ZK_API_BEGIN(ctx) ... ZK_API_END (ctx.hpp: device guard + exception barrier).
Idempotent: bodies that already start with ZK_API_BEGIN are left alone.   python tools/wrap_abi.py [--check]"""
import glob, os, re, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zk-mpc_amd", "csrc")


def match_brace(s, i):
    """s[i] == '{' -> index of the matching '}', skipping strings, chars and comments."""
    depth, n = 0, len(s)
    while i < n:
        c = s[i]
        if s.startswith("//", i):
            i = s.index("\n", i)
            continue
        if s.startswith("/*", i):
            i = s.index("*/", i) + 2
            continue
        if c == '"' or c == "'":
            j = i + 1
            while s[j] != c:
                j += 2 if s[j] == "\\" else 1
            i = j + 1
            continue
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ValueError("unbalanced braces")


def entries(s):
    """(name, has_ctx, body_open, body_close) of every extern "C" int definition."""
    out = []
    for m in re.finditer(r'^extern "C" int (\w+)\(', s, re.M):
        i, depth = m.end() - 1, 0
        while True:                                   # the parameter list
            if s[i] == "(":
                depth += 1
            elif s[i] == ")":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        params = s[m.end():i]
        j = i + 1
        while s[j] in " \t\n":
            j += 1
        if s[j] != "{":
            continue                                  # a declaration
        out.append((m.group(1), bool(re.search(r"\bzk_ctx\s*\*\s*ctx\b", params)), j, match_brace(s, j)))
    return out


def main():
    check = "--check" in sys.argv
    bad = []
    for path in sorted(glob.glob(os.path.join(ROOT, "*.hip"))):
        s = open(path).read()
        changed = False
        for name, has_ctx, a, b in reversed(entries(s)):
            body = s[a + 1:b]
            if body.lstrip().startswith("ZK_API_BEGIN"):
                continue
            if check:
                bad.append("%s: %s" % (os.path.basename(path), name))
                continue
            begin = "ZK_API_BEGIN(ctx)" if has_ctx else "ZK_API_BEGIN_NOCTX"
            if "\n" in body.strip("\n") or body.startswith("\n"):
                new = "\n    " + begin + body.rstrip(" \n") + "\n    ZK_API_END\n"
            else:
                new = " " + begin + " " + body.strip() + " ZK_API_END "
            s = s[:a + 1] + new + s[b:]
            changed = True
        if changed:
            open(path, "w").write(s)
    if check:
        print("\n".join(bad) if bad else "every extern \"C\" int entry point is guarded")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
