#!/usr/bin/env python3
"""Diffs the Rust FFI block (rust/cftrt-sys/src/lib.rs) against include/cftrt.h.

There is no Rust toolchain in the build image, so the crate cannot be compiled; this script is what keeps its text
honest.  It parses both files and reports
  * functions declared on one side only, differing argument counts, argument or return types;
  * #[repr(C)] structs whose fields differ in name, order or type, and their sizes under the C layout rules;
  * #define constants of the header with a different value (or missing) among the Rust `pub const`s.
Exit status 0 = in step.  Usage: check_rust_ffi.py [header] [lib.rs]"""
import os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "include", "cftrt.h")
RS = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "rust", "cftrt-sys", "src", "lib.rs")

C_SCALAR = {"int": "c_int", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "double": "f64",
            "float": "f32", "uint8_t": "u8", "unsigned char": "c_uchar", "char": "c_char", "void": "c_void"}
SIZE = {"c_int": 4, "i32": 4, "u32": 4, "i64": 8, "u64": 8, "f64": 8, "f32": 4, "u8": 1, "c_uchar": 1, "c_char": 1}


def strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def c_type_to_rust(t):
    """'const cftrt_camera*' -> '*const cftrt_camera'; 'cftrt_ctx**' -> '*mut *mut cftrt_ctx'; 'uint32_t' -> 'u32'."""
    t = " ".join(t.replace("*", " * ").split())
    stars = t.count("*")
    base = t.replace("*", " ").split()
    const = "const" in base
    base = " ".join(w for w in base if w not in ("const", "struct"))
    r = C_SCALAR.get(base, base)
    for i in range(stars):
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def parse_header(path):
    s = strip_c_comments(open(path).read())
    consts = {}
    for m in re.finditer(r"^#define\s+(CFTRT_\w+)\s+(.+?)\s*$", s, flags=re.M):
        v = m.group(2).strip()
        v = re.sub(r"(\d+)u\b", r"\1", v)
        try:
            consts[m.group(1)] = int(eval(v, {"__builtins__": {}}, {}))
        except Exception:
            pass
    structs = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", s, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ty, names = decl.rsplit(" ", 1)[0], None
            # "double p0, p1" / "double color[3]" / "uint64_t rays_per_level[CFTRT_MAX_LEVELS]"
            mm = re.match(r"((?:unsigned )?\w+)\s+(.*)", decl)
            ty, names = mm.group(1), mm.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                am = re.match(r"(\w+)\[(\w+)\]", nm)
                if am:
                    n = am.group(2)
                    n = consts[n] if n in consts else int(n)
                    fields.append((am.group(1), "[%s; %d]" % (C_SCALAR[ty], n)))
                else:
                    fields.append((nm, C_SCALAR[ty]))
        structs[m.group(1)] = fields
    body = re.sub(r"typedef struct \w+ \{.*?\} \w+;", " ", s, flags=re.S)
    funcs = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[\w ]+?\**)\s*(cftrt_\w+)\s*\(([^)]*)\)\s*;", body):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        if ret.startswith("typedef"):
            continue
        al = []
        for a in args.split(","):
            a = " ".join(a.split())
            if not a or a == "void":
                continue
            mm = re.match(r"(.*?)(\w+)$", a)
            al.append((mm.group(2), c_type_to_rust(mm.group(1))))
        funcs[name] = (None if ret == "void" else c_type_to_rust(ret), al)
    return consts, structs, funcs


def parse_rust(path):
    s = re.sub(r"//.*", "", open(path).read())
    consts = {}
    for m in re.finditer(r"pub const (\w+): \w+ = ([^;]+);", s):
        v = re.sub(r"(\d+)(?:u32|u64|usize|i32)\b", r"\1", m.group(2))
        try:
            consts[m.group(1)] = int(eval(v, {"__builtins__": {}}, {}))
        except Exception:
            pass
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^)]*\)\])?\s*pub struct (\w+)\s*\{(.*?)\}", s, flags=re.S):
        fields = []
        for f in m.group(2).split(","):
            f = " ".join(f.split())
            if not f:
                continue
            mm = re.match(r"(?:pub )?(\w+): (.+)", f)
            fields.append((mm.group(1), mm.group(2).strip()))
        structs[m.group(1)] = fields
    funcs = {}
    ext = re.search(r'extern "C" \{(.*)\}', s, flags=re.S).group(1)
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext, flags=re.S):
        al = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            if not a:
                continue
            nm, ty = a.split(":", 1)
            al.append((nm.strip(), ty.strip()))
        funcs[m.group(1)] = (m.group(3).strip() if m.group(3) else None, al)
    return consts, structs, funcs


def c_layout_size(fields):
    off, align = 0, 1
    for _, ty in fields:
        m = re.match(r"\[(\w+); (\d+)\]", ty)
        el, n = (m.group(1), int(m.group(2))) if m else (ty, 1)
        sz = SIZE[el]
        align = max(align, sz)
        off = (off + sz - 1) // sz * sz + sz * n
    return (off + align - 1) // align * align


def main():
    hc, hs, hf = parse_header(HDR)
    rc, rs_, rf = parse_rust(RS)
    errs = []
    for name in sorted(set(hf) | set(rf)):
        if name not in rf:
            errs.append(f"function {name}: in the header, not in the Rust block")
        elif name not in hf:
            errs.append(f"function {name}: in the Rust block, not in the header")
        else:
            (hr, ha), (rr, ra) = hf[name], rf[name]
            if hr != rr:
                errs.append(f"function {name}: return type {hr} (header) vs {rr} (Rust)")
            if len(ha) != len(ra):
                errs.append(f"function {name}: {len(ha)} arguments (header) vs {len(ra)} (Rust)")
            else:
                for i, ((hn, ht), (rn, rt)) in enumerate(zip(ha, ra)):
                    if ht != rt:
                        errs.append(f"function {name}: argument {i} `{hn}`: {ht} (header) vs {rt} (Rust)")
    opaque = {"cftrt_ctx", "cftrt_scene"}
    for name in sorted(set(hs) | (set(rs_) - opaque)):
        if name not in rs_:
            errs.append(f"struct {name}: in the header, not in the Rust block")
        elif name not in hs:
            errs.append(f"struct {name}: in the Rust block, not in the header")
        else:
            if hs[name] != rs_[name]:
                errs.append(f"struct {name}: fields differ\n    header: {hs[name]}\n    Rust:   {rs_[name]}")
            else:
                print(f"struct {name}: {len(hs[name])} fields, {c_layout_size(hs[name])} bytes on both sides")
    for name, v in sorted(hc.items()):
        if name == "CFTRT_H":
            continue
        if name not in rc:
            errs.append(f"constant {name} = {v}: not among the Rust consts")
        elif rc[name] != v:
            errs.append(f"constant {name}: {v} (header) vs {rc[name]} (Rust)")
    print(f"{len(hf)} functions in the header, {len(rf)} in the Rust block; {len(hc) - ('CFTRT_H' in hc)} constants")
    for e in errs:
        print("MISMATCH:", e)
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
