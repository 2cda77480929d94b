"""Static conformance of the cgo package go/cometgpu/ (no Go toolchain in this image: nothing has ever parsed these files).

1. every `C.comet_*` call names a function include/comet_gpu.h declares and passes as many arguments as its prototype has;
   every `C.COMET_*` constant and `C.comet_*` type is defined in the header;
2. every method of the reference's interfaces — VectorIndex (index.go:32-63), VectorSearch (index_search.go:141-279), TextIndex
   (index.go:65-81), TextSearch (index_search.go:306-430), transcribed below — has a receiver in the package on the type the
   package asserts to implement it (`var _ comet.X = (*T)(nil)`);
3. a Go pointer stored inside a struct that is handed to C is pinned (cgo's pointer-passing rules): every `filter_ids` assignment
   is preceded by a runtime.Pinner Pin of the same slice.
"""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
GO = sorted((ROOT / "go" / "cometgpu").glob("*.go"))
HEADER = (ROOT / "include" / "comet_gpu.h").read_text()

# the reference's interfaces, method name -> number of parameters (variadic counts as one)
REFERENCE_INTERFACES = {
    "VectorIndex": {"Train": 1, "Add": 1, "Remove": 1, "Flush": 0, "NewSearch": 0, "Dimensions": 0, "DistanceKind": 0, "Kind": 0, "Trained": 0,
                    "WriteTo": 1, "ReadFrom": 1},                                                            # index.go:32-63 (+ io.WriterTo / io.ReaderFrom)
    "VectorSearch": {"WithQuery": 1, "WithNode": 1, "WithK": 1, "WithNProbes": 1, "WithEfSearch": 1, "WithThreshold": 1, "WithScoreAggregation": 1,
                     "WithCutoff": 1, "WithDocumentIDs": 1, "WithReranker": 1, "Execute": 0},                # index_search.go:141-279
    "TextIndex": {"Add": 2, "Remove": 1, "NewSearch": 0, "Flush": 0, "WriteTo": 1, "ReadFrom": 1},           # index.go:65-81
    "TextSearch": {"WithQuery": 1, "WithNode": 1, "WithK": 1, "WithScoreAggregation": 1, "WithCutoff": 1, "WithDocumentIDs": 1, "Execute": 0},   # index_search.go:306-430
}


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", lambda m: " " * len(m.group(0)) if "\n" not in m.group(0) else "\n" * m.group(0).count("\n"), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def header_prototypes():
    """function name -> parameter count, from the COMET_API declarations"""
    protos = {}
    for m in re.finditer(r"COMET_API\s+[^;(]*?\b(comet_\w+)\s*\(([^;]*?)\)\s*;", strip_comments(HEADER), flags=re.S):
        params = m.group(2).strip()
        if params in ("", "void"):
            n = 0
        else:
            depth, n = 0, 1
            for ch in params:          # function-pointer parameters carry their own commas
                depth += ch == "("
                depth -= ch == ")"
                n += ch == "," and depth == 0
        protos[m.group(1)] = n
    return protos


def split_args(s: str):
    """arguments of a call whose text starts right after the opening parenthesis; returns (args, rest)"""
    depth, cur, args = 0, "", []
    in_str = None
    for i, ch in enumerate(s):
        if in_str:
            cur += ch
            if ch == in_str and s[i - 1] != "\\":
                in_str = None
            continue
        if ch in "\"`'":
            in_str = ch; cur += ch; continue
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            if depth == 0:
                if cur.strip():
                    args.append(cur.strip())
                return args, s[i + 1:]
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur.strip()); cur = ""
        else:
            cur += ch
    raise AssertionError("unbalanced call")


def go_sources():
    return {f.name: strip_comments(f.read_text()) for f in GO if not f.name.endswith("_test.go")}


def test_every_c_call_matches_a_prototype():
    protos = header_prototypes()
    assert len(protos) >= 80, "header parse lost its declarations"
    types = set(re.findall(r"\b(comet_\w+)\b", strip_comments(HEADER))) - set(protos)
    consts = set(re.findall(r"\b(COMET_\w+)\b", strip_comments(HEADER)))
    calls = 0
    for name, src in go_sources().items():
        for m in re.finditer(r"\bC\.(comet_\w+)\s*(\()?", src):
            fn = m.group(1)
            if m.group(2) is None:      # a type (C.comet_search_params{...}, *C.comet_index) or a function value
                assert fn in types or fn in protos, f"{name}: C.{fn} is not declared in include/comet_gpu.h"
                continue
            if fn in types and fn not in protos:
                continue                # conversion to a C type: C.comet_x(...)
            assert fn in protos, f"{name}: C.{fn}() is not declared in include/comet_gpu.h"
            args, _ = split_args(src[m.end():])
            assert len(args) == protos[fn], f"{name}: C.{fn} called with {len(args)} arguments, the header declares {protos[fn]}: {args}"
            calls += 1
        for m in re.finditer(r"\bC\.(COMET_\w+)\b", src):
            assert m.group(1) in consts, f"{name}: C.{m.group(1)} is not defined in include/comet_gpu.h"
    assert calls >= 40, f"only {calls} calls found: the scanner lost the package"


def receivers():
    """type -> {method: parameter count} over the package"""
    out = {}
    for src in go_sources().values():
        for m in re.finditer(r"^func\s*\(\s*\w+\s+\*?(\w+)\s*\)\s*(\w+)\s*\(", src, flags=re.M):
            args, _ = split_args(src[m.end():])
            # `a, b T` declares two parameters in one argument text
            n = 0
            for a in args:
                n += 1
            out.setdefault(m.group(1), {})[m.group(2)] = (n, args)
    return out


def param_count(args):
    """Go parameter lists group names: `ids []uint32, vecs [][]float32` is two, `a, b int` is two as well"""
    return len(args)


def test_reference_interfaces_are_implemented():
    srcs = "\n".join(go_sources().values())
    asserted = dict((iface, typ) for iface, typ in re.findall(r"var\s+_\s+comet\.(\w+)\s*=\s*\(\*(\w+)\)\(nil\)", srcs))
    recv = receivers()
    for iface, methods in REFERENCE_INTERFACES.items():
        assert iface in asserted, f"the package never asserts to implement comet.{iface}"
        have = recv.get(asserted[iface], {})
        for meth, nparams in methods.items():
            assert meth in have, f"{asserted[iface]} lacks {iface}.{meth}"
            assert param_count(have[meth][1]) == nparams, f"{asserted[iface]}.{meth} takes {have[meth][1]}, comet.{iface}.{meth} takes {nparams} parameter(s)"


def test_go_pointers_inside_c_structs_are_pinned():
    for name, src in go_sources().items():
        for m in re.finditer(r"(\w+)\.filter_ids\s*=\s*\(\*C\.uint32_t\)\(&(\w[\w.]*)\[0\]\)", src):
            before = src[:m.start()]
            assert re.search(r"\.Pin\(&" + re.escape(m.group(2)) + r"\[0\]\)", before[-400:]), f"{name}: filter_ids takes &{m.group(2)}[0] without pinning it"
            assert "runtime" in re.search(r"import\s*\((.*?)\)", src, flags=re.S).group(1), f"{name}: runtime is not imported"
