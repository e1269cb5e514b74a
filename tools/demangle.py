"""Kernel names for the rocprofv3 summaries.  rocprofv3 is run with -M (mangled names): its own demangler does not know the bf16
type code `DF16b` (it leaves such names mangled or garbles them into `bool _Accum`).  Here `DF16b` is rewritten to the half code
`Dh` (never used by this library), binutils' c++filt demangles, and `half` is written back as `__bf16` — the spelling bench.py and
ops._Timed use."""
import re
import subprocess

_cache = {}


def demangle(names):
    todo = sorted({n for n in names if n.startswith("_Z") and n not in _cache})
    if todo:
        try:
            out = subprocess.run(["c++filt"], input="\n".join(n.replace("DF16b", "Dh") for n in todo), capture_output=True, text=True, check=True).stdout.split("\n")
        except (OSError, subprocess.CalledProcessError):
            out = todo
        for n, d in zip(todo, out):
            _cache[n] = d.replace("half", "__bf16")
    return [_cache.get(n, n) for n in names]


def short(name):
    """demangled signature -> `kernel<template args>` (no `void`, no parameter list)"""
    name = re.sub(r"^void\s+", "", name)
    depth, i = 0, 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i].strip()
    return name.strip()


def norm(name):
    return short(demangle([name])[0])
