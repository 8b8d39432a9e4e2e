"""The Rust shim (kzero_amd/rust/*.rs) cannot be compiled here (no cargo in the image), so it is at least kept
consistent as text: every prototype of include/kz_hip.h is bound in hip.rs with the same name, arity, pointer-ness and
scalar types (as test_abi.py does for capi.py); the `#[repr(C)]` info struct and the constants match; and
server_hip.rs only names items hip.rs defines and calls them with the arity they have."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(REPO, "include", "kz_hip.h")).read()
HIP_RS = open(os.path.join(REPO, "kzero_amd", "rust", "hip.rs")).read()
SERVER_RS = open(os.path.join(REPO, "kzero_amd", "rust", "server_hip.rs")).read()

C_TO_RUST = {
    "int": "c_int", "size_t": "usize", "const char *": "*const c_char", "char *": "*mut c_char",
    "const float *": "*const f32", "float *": "*mut f32", "const float **": "*mut *const f32",
    "const uint8_t *": "*const u8", "const int64_t *": "*const i64", "const int32_t *": "*const i32",
    "int *": "*mut c_int", "double *": "*mut f64", "int64_t *": "*mut i64",
    "kz_model **": "*mut *mut c_void", "const kz_model *": "*const c_void", "kz_model *": "*mut c_void",
    "kz_engine **": "*mut *mut c_void", "const kz_engine *": "*const c_void", "kz_engine *": "*mut c_void",
    "const void *": "*const c_void", "void *": "*mut c_void", "void **": "*mut *mut c_void",
    "kz_model_info *": "*mut KzModelInfo", "kz_path_plan *": "*mut KzPathPlan",
}
C_RET = {"int": "c_int", "void": None, "const char *": "*const c_char"}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def header_prototypes():
    body = strip_c_comments(HEADER)
    body = "\n".join(ln for ln in body.splitlines() if not ln.lstrip().startswith("#"))
    protos = {}
    for m in re.finditer(r"(?:^|;|\})\s*((?:const\s+)?\w+\s*\**)\s*(kz_\w+)\s*\(([^)]*)\)\s*(?=;)", body, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        ret = re.sub(r"\s+", " ", ret).replace(" *", " *").strip()
        ret = re.sub(r"\s*\*", " *", ret) if "*" in ret else ret
        params = []
        if args.strip() and args.strip() != "void":
            for a in args.split(","):
                a = re.sub(r"\s+", " ", a).strip()
                mm = re.match(r"(.*?)(\w+)$", a)  # type, then the parameter name
                ctype = mm.group(1).strip()
                ctype = re.sub(r"\s*\*", "*", ctype)
                stars = ctype.count("*")
                ctype = ctype.replace("*", "").strip() + (" " + "*" * stars if stars else "")
                params.append(ctype)
        protos[name] = (ret, params)
    return protos


def rust_externs():
    block = re.search(r'extern "C" \{(.*?)\n\}', HIP_RS, flags=re.S).group(1)
    block = re.sub(r"//[^\n]*", "", block)
    fns = {}
    for m in re.finditer(r"fn\s+(kz_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(3)
        params = [re.sub(r"\s+", " ", a.split(":", 1)[1]).strip() for a in args.split(",") if a.strip()]
        fns[name] = (ret.strip() if ret else None, params)
    return fns


def test_every_header_function_is_bound_in_hip_rs_with_the_same_signature():
    protos, fns = header_prototypes(), rust_externs()
    assert len(protos) >= 30, sorted(protos)
    assert set(protos) == set(fns), f"header only: {sorted(set(protos) - set(fns))}; hip.rs only: {sorted(set(fns) - set(protos))}"
    for name, (ret, params) in protos.items():
        r_ret, r_params = fns[name]
        assert C_RET[ret] == r_ret, f"{name}: return {ret} vs {r_ret}"
        assert len(params) == len(r_params), f"{name}: arity {len(params)} vs {len(r_params)}"
        for i, (c, r) in enumerate(zip(params, r_params)):
            assert c in C_TO_RUST, f"{name} arg {i}: unmapped C type '{c}'"
            assert C_TO_RUST[c] == r, f"{name} arg {i}: {c} should be {C_TO_RUST[c]}, hip.rs has {r}"


def test_capi_py_binds_the_same_set():
    from kzero_amd import capi
    assert set(capi.SIGNATURES) == set(header_prototypes())


def test_model_info_struct_and_constants_match():
    body = strip_c_comments(HEADER)
    c_fields = re.search(r"typedef struct kz_model_info \{(.*?)\} kz_model_info;", body, flags=re.S).group(1)
    c_fields = [(t, n) for t, n in re.findall(r"(int32_t|int64_t|double)\s+(\w+);", c_fields)]
    r_struct = re.search(r"pub struct KzModelInfo \{(.*?)\}", HIP_RS, flags=re.S).group(1)
    r_fields = re.findall(r"pub (\w+): (\w+),", r_struct)
    tmap = {"int32_t": "i32", "int64_t": "i64", "double": "f64"}
    assert [(n, tmap[t]) for t, n in c_fields] == r_fields
    assert "#[repr(C)]" in HIP_RS.split("pub struct KzModelInfo")[0][-80:]
    for name in ("KZ_DTYPE_F32", "KZ_DTYPE_F16", "KZ_DTYPE_F32_SPLIT16", "KZ_ENGINE_SLOTS"):
        c_val = re.search(rf"#define {name} (\d+)", HEADER).group(1)
        r_val = re.search(rf"pub const {name}: \w+ = (\d+);", HIP_RS).group(1)
        assert c_val == r_val, name


def _code_only(rs):
    return "\n".join(ln for ln in rs.splitlines() if not ln.lstrip().startswith("//"))


def test_server_hip_rs_is_self_consistent():
    code = _code_only(SERVER_RS)
    # the kn-cuda-sys device type is gone from the seam: only comments may mention it
    assert "CudaDevice" not in code and "KZ_DTYPE_F16" not in code
    # everything imported from kz_core::network::hip exists there as a public item
    imported = re.search(r"use kz_core::network::hip::\{([^}]*)\};", code).group(1)
    for item in [i.strip() for i in imported.split(",")]:
        assert re.search(rf"pub (struct|enum|fn|const) {item}\b", HIP_RS), f"hip.rs does not define {item}"
    # HipNetwork::new: same number of arguments at the call and at the definition, device and dtype typed as the seam's
    defn = re.search(r"pub fn new\(([^)]*)\) -> Self", HIP_RS).group(1)
    def_args = [a.strip() for a in defn.split(",") if a.strip()]
    call = re.search(r"HipNetwork::new\(([^)]*)\)", code).group(1)
    assert len([a for a in call.split(",") if a.strip()]) == len(def_args) == 5
    assert "device: HipDevice" in defn and "dtype: HipDtype" in defn
    # the trait items this impl provides are the ones the edited trait (documented at the top of the file) declares
    for item in ("type G = HipModel;", "type Device = HipDevice;", "fn devices(indices: &[i32]) -> Vec<HipDevice>",
                 "fn spawn_device_threads<'s>(", "fn load_graph(&self, path: &str, mapper: M, _: &StartupSettings) -> HipModel"):
        assert item in code, item
    # the patch adds no copy of reference code: the executor loop, the job channel, the generator pool and
    # the symmetry wrapper stay in the shared helper of server_alphazero.rs — this file only names the helper
    for reference_only in ("batched_executor_loop", "RunCondition", "ThreadPoolBuilder", "job_pair", "RandomSymmetryNetwork",
                           "generator_alphazero_main", "flume::bounded", "Evals::new"):
        assert reference_only not in code, reference_only
    assert "spawn_alphazero_device_threads(" in code
    assert len([ln for ln in code.splitlines() if ln.strip()]) < 70
    # the default arithmetic is the <= 1e-4 path, f16 is opt-in
    assert re.search(r'Err\(_\) \| Ok\("parity"\)[^\n]*=> HipDtype::Parity', HIP_RS)
    assert "HipDtype::Parity =>" in HIP_RS and "KZ_DTYPE_F32_SPLIT16" in HIP_RS


def test_integration_md_lists_every_edited_reference_line():
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    for cite in ("server_alphazero.rs:39-123", "load_network", "server.rs:16", "server.rs:47-53", "server.rs:105", "server.rs:207", "server.rs:249", "server.rs:293",
                 "server.rs:306", "server_alphazero.rs:7", "server_alphazero.rs:35", "server_muzero.rs:26",
                 "type Device", "KZ_HIP_DTYPE", "Legacy three-output graphs are taken"):
        assert cite in text, f"INTEGRATION.md does not mention {cite}"


def _strip_rust(text):
    """Rust source without comments, string / char literals and lifetimes: what is left must have balanced delimiters."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            i = text.find("\n", i) if "\n" in text[i:] else n
        elif text.startswith("/*", i):
            depth, i = 1, i + 2
            while i < n and depth:
                if text.startswith("/*", i):
                    depth, i = depth + 1, i + 2
                elif text.startswith("*/", i):
                    depth, i = depth - 1, i + 2
                else:
                    i += 1
        elif c == '"':
            i += 1
            while i < n and text[i] != '"':
                i += 2 if text[i] == "\\" else 1
            i += 1
        elif c == "'":
            m = re.match(r"'(\\.|[^\\'])'", text[i:])  # a char literal; otherwise a lifetime ('static, 's)
            i += m.end() if m else 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def test_rust_sources_are_lexically_well_formed():
    """No compiler here, so at least: every delimiter of the three Rust files closes in the right order (comments, string
    and char literals and lifetimes aside), and no statement-level typo like a doubled `;;` or an `fn` without a body slipped in."""
    rust_dir = os.path.join(REPO, "kzero_amd", "rust")
    pairs = {")": "(", "]": "[", "}": "{"}
    for name in sorted(os.listdir(rust_dir)):
        if not name.endswith(".rs"):
            continue
        code = _strip_rust(open(os.path.join(rust_dir, name)).read())
        stack = []
        for pos, c in enumerate(code):
            if c in "([{":
                stack.append((c, pos))
            elif c in ")]}":
                assert stack and stack[-1][0] == pairs[c], f"{name}: unbalanced '{c}' near ...{code[max(0, pos - 60):pos + 1]!r}"
                stack.pop()
        assert not stack, f"{name}: unclosed '{stack[-1][0]}' near ...{code[stack[-1][1]:stack[-1][1] + 60]!r}"
        assert ";;" not in code, name
        for m in re.finditer(r"\bfn\s+\w+[^;{]*([;{])", code):  # a fn outside a trait / extern block has a body
            if m.group(1) == ";":
                before = code[:m.start()]
                assert before.rfind('extern') > before.rfind("}\n\n") or "trait" in before[before.rfind("\n\n"):], f"{name}: fn without a body: {m.group(0)[:60]}"
