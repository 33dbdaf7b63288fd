"""The Rust FFI block (rust/cftrt-sys/src/lib.rs) cannot be compiled here (no rustc in the image); it is kept in
step with include/cftrt.h by scripts/check_rust_ffi.py, and the checker itself is checked: it must agree with the
ctypes mirror on the struct sizes and must notice a drifted signature, field or constant."""
import ctypes, importlib, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "check_rust_ffi.py")
RS = os.path.join(ROOT, "rust", "cftrt-sys", "src", "lib.rs")


def _run(*args):
    return subprocess.run([sys.executable, SCRIPT, *args], capture_output=True, text=True)


def test_rust_block_matches_header():
    r = _run()
    assert r.returncode == 0, r.stdout + r.stderr
    sizes = {m.group(1): int(m.group(2)) for m in re.finditer(r"struct (\w+): \d+ fields, (\d+) bytes", r.stdout)}
    lib = importlib.import_module("ray-tracing_amd._lib")
    for name in ("cftrt_material", "cftrt_light", "cftrt_camera", "cftrt_stats"):
        assert sizes[name] == ctypes.sizeof(getattr(lib, name)), name
    # every exported symbol of the ctypes mirror is declared in the Rust block too
    text = open(RS).read()
    for sym, _ret, args in lib.SYMBOLS:
        m = re.search(r"pub fn %s\s*\((.*?)\)" % sym, text, flags=re.S)
        assert m, sym
        assert len([a for a in m.group(1).split(",") if a.strip()]) == len(args), sym   # and as many arguments


def test_checker_notices_drift(tmp_path):
    text = open(RS).read()
    cases = [
        ("out_outward: *mut i32,\n", ""),                                  # an argument dropped
        ("pub fn cftrt_ctx_set_finish(ctx: *mut cftrt_ctx, on: c_int, below: u32)", "pub fn cftrt_ctx_set_finish(ctx: *mut cftrt_ctx, on: c_int, below: u64)"),
        ("    pub passes: u32,\n", ""),                                    # a struct field dropped
        ("pub const CFTRT_COMM_ID_BYTES: usize = 128;", "pub const CFTRT_COMM_ID_BYTES: usize = 64;"),
        ("    pub fn cftrt_comm_destroy(ctx: *mut cftrt_ctx) -> c_int;\n", ""),   # a function dropped
    ]
    for old, new in cases:
        assert text.count(old) == 1, old
        p = tmp_path / "lib.rs"
        p.write_text(text.replace(old, new))
        r = _run(os.path.join(ROOT, "include", "cftrt.h"), str(p))
        assert r.returncode == 1 and "MISMATCH" in r.stdout, (old, r.stdout)


def test_facade_and_examples_use_declared_names_only():
    """rust/facade/gpu.rs and rust/examples/*.rs are text too: every cftrt_* / CFTRT_* name they use must be one the FFI
    crate declares (a function, a struct or a constant), and every World method the example text calls must be one the
    façade defines."""
    decl = open(RS).read()
    declared = set(re.findall(r"pub fn (cftrt_\w+)", decl)) | set(re.findall(r"pub struct (cftrt_\w+)", decl)) | \
        set(re.findall(r"pub const (CFTRT_\w+)", decl))
    facade = open(os.path.join(ROOT, "rust", "facade", "gpu.rs")).read()
    files = [os.path.join(ROOT, "rust", "facade", "gpu.rs")]
    exdir = os.path.join(ROOT, "rust", "examples")
    files += [os.path.join(exdir, f) for f in sorted(os.listdir(exdir)) if f.endswith(".rs")]
    assert len(files) >= 2
    for f in files:
        code = re.sub(r"//.*", "", open(f).read())
        used = set(re.findall(r"\b(cftrt_\w+|CFTRT_\w+)\b", code)) - {"cftrt_sys"}   # (the crate's own name)
        assert used <= declared, (f, sorted(used - declared))
        if "examples" in f:
            for m in set(re.findall(r"world\.(\w+)\(", code)):
                assert re.search(r"pub fn %s\b" % m, facade), (f, m)
