#!/usr/bin/env python3
"""Writes patches/xllm-use-mi355.patch: the reference-side binding of the MI355X backend, every hunk written out.

The reference selects its backend at compile time. Two kinds of gate exist:

 (K) operator gates -- the `#if defined(USE_<BACKEND>)` chains of xllm/core/kernels/ops_api.cpp, the attention class picked by
     layers/common/attention.h, the three host files that include a kernel header directly. Here a USE_MI355 build takes its
     OWN branch: `kernels/mi355/mi355_ops_api.h` (= shim/mi355_ops_api.h, the argument lists of cuda_ops_api.h / dcu_ops_api.h)
     and `layers/mi355/attention.h` (= shim/mi355_attention.h).
 (H) host gates -- everything else: platform (device, stream, VMM), batch builder, KV-cache shape, worker, graph executor,
     sampler, process groups ... The DCU branch of each is plain HIP / ROCm-torch host code (c10::hip, hipMemCreate, RCCL through
     c10d::ProcessGroupNCCL), nothing vendor-specific, so the rule is ONE rule: **every host gate that accepts USE_DCU accepts
     USE_MI355** (`defined(USE_DCU)` -> `defined(USE_DCU) || defined(USE_MI355)`, `!defined(USE_DCU)` ->
     `!defined(USE_DCU) && !defined(USE_MI355)`), in the C++ sources and in the CMake lists that pick the DCU host files
     (runtime/dcu_graph_executor_impl.*, platform/dcu/*). `Platform::type_str()` therefore stays "dcu" (device names `dcu:N`,
     executor registry key "dcu"): to the host code an MI355X is a device of the HIP family.

The script discovers every non-test file of the reference that mentions USE_DCU and sorts it into one of: rule (H), a
specific edit (K), or REPLACED (kernel sources the USE_MI355 build does not compile -- kernels/cuda/*, served by
libxllm_mi355.so). tests/test_host_abi.py walks all of them: a preprocessor run of the patched file with only USE_MI355
defined must leave exactly the lines the original leaves with only USE_DCU defined, except the differences listed in
EXPECTED_DIFF below. Run it where /root/reference exists; the patch it writes is committed (context lines are the
reference's, as in any patch; no reference source is copied into this repository otherwise).

  python tools/make_reference_patch.py            # writes the patch
  python tools/make_reference_patch.py --gates    # prints the gate table of INTEGRATION.md section 3.6
"""
import difflib
import os
import re
import sys

REF = os.environ.get("XLLM_REFERENCE", "/root/reference")       # or the first positional argument when run as a script
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "patches", "xllm-use-mi355.patch")
MI355 = " || defined(USE_MI355)"

# kernel sources a USE_MI355 build does not compile: the hipified CUDA kernels of the DCU build (kernels/dcu/CMakeLists.txt
# globs kernels/cuda/*.cu) and the two operator headers, all replaced by kernels/mi355/mi355_ops_api.{h,cpp} over
# libxllm_mi355.so. The walk test asserts that no host file includes any of them under USE_MI355.
REPLACED_PREFIXES = ("xllm/core/kernels/cuda/",)

# files whose USE_DCU mentions are not a compile-time gate of the hot path's host code
NOT_A_GATE = {
    "setup.py": "build driver: gets a `--device mi355` arm of its own (specific edit)",
}


# ------------------------------------------------------------------------------------------------ rule (H), C++ sources
def _directives(src: str):
    """yields (first_line, last_line_exclusive) of every #if / #elif directive, continuation lines included"""
    lines = src.split("\n")
    i = 0
    while i < len(lines):
        if re.match(r"[ \t]*#[ \t]*(?:if|elif)\b", lines[i]):
            j = i
            while lines[j].rstrip().endswith("\\") and j + 1 < len(lines):
                j += 1
            yield i, j + 1
            i = j + 1
        else:
            i += 1


def gate_everywhere(src: str) -> str:
    """rule (H): every preprocessor condition that mentions USE_DCU treats USE_MI355 the same way"""
    lines = src.split("\n")
    for a, b in _directives(src):
        text = " ".join(l.rstrip("\\") for l in lines[a:b])
        if "USE_DCU" not in text or "USE_MI355" in text:
            continue
        expr = re.sub(r"/\*.*?\*/|//.*", "", text)
        negated = len(re.findall(r"!\s*defined\s*\(", expr))
        total = len(re.findall(r"defined\s*\(", expr))
        pure_or = "&&" not in expr and negated == 0
        pure_nand = "||" not in expr and negated == total
        for k in range(a, b):
            if "USE_DCU" not in lines[k]:
                continue
            if pure_or:
                lines[k] = lines[k].replace("defined(USE_DCU)", "defined(USE_DCU) || defined(USE_MI355)")
            elif pure_nand:
                lines[k] = re.sub(r"!\s*defined\(USE_DCU\)", "!defined(USE_DCU) && !defined(USE_MI355)", lines[k])
            else:
                lines[k] = re.sub(r"!\s*defined\(USE_DCU\)", "!(defined(USE_DCU) || defined(USE_MI355))", lines[k])
                lines[k] = re.sub(r"(?<![!(])\bdefined\(USE_DCU\)(?! \|\| defined\(USE_MI355\))",
                                  "(defined(USE_DCU) || defined(USE_MI355))", lines[k])
    return "\n".join(lines)


# ------------------------------------------------------------------------------------------------ rule (H), CMake lists
def cmake_gate_everywhere(src: str) -> str:
    out = []
    for line in src.split("\n"):
        if "USE_DCU" in line and "USE_MI355" not in line and not line.lstrip().startswith(("#", "option(", "add_definitions(")):
            line = re.sub(r"\b(if|elseif)\(USE_DCU\)", r"\1(USE_DCU OR USE_MI355)", line)
            line = re.sub(r"\b(if|elseif)\(USE_DCU AND ", r"\1((USE_DCU OR USE_MI355) AND ", line)
            line = re.sub(r"\bOR USE_DCU\)", "OR USE_DCU OR USE_MI355)", line)
            line = line.replace("$<$<BOOL:${USE_DCU}>:", "$<$<OR:$<BOOL:${USE_DCU}>,$<BOOL:${USE_MI355}>>:")
            line = line.replace(",$<BOOL:${USE_DCU}>>", ",$<BOOL:${USE_DCU}>,$<BOOL:${USE_MI355}>>")
        out.append(line)
    return "\n".join(out)


# ------------------------------------------------------------------------------------------------ specific edits (K)
MI355_OPS_INCLUDE = ('#include "kernels/mi355/mi355_ops_api.h"\n'
                     'namespace xllm::kernel {\nnamespace cuda = mi355;\n}  // namespace xllm::kernel\n')


def edit_ops_api(src: str) -> str:
    # (1) header + namespace aliases instead of the CUDA / DCU headers
    src = src.replace('''#elif defined(USE_DCU)
#include "cuda/cuda_ops_api.h"
#include "dcu/dcu_ops_api.h"
#endif
''', '''#elif defined(USE_DCU)
#include "cuda/cuda_ops_api.h"
#include "dcu/dcu_ops_api.h"
#elif defined(USE_MI355)
// AMD Instinct MI355X (gfx950): xllm::kernel::mi355::* has the argument lists of the cuda:: / dcu:: functions the DCU
// branches below call, so those branches are reused verbatim through two namespace aliases
#include "mi355/mi355_ops_api.h"
namespace xllm::kernel {
namespace cuda = mi355;
namespace dcu = mi355;
}  // namespace xllm::kernel
#endif
''', 1)
    head, sep, rest = src.partition("#include <numeric>")
    rest = gate_everywhere(rest)
    # (2) the fp8 operators exist on CUDA only in the reference; the MI355 backend implements them
    for fn in ("fp8_scaled_quantize(\n    Fp8ScaledQuantizeParams& params)", "fp8_scaled_matmul(Fp8ScaledMatmulParams& params)",
               "static_scaled_fp8_quant(StaticScaledFp8QuantParams& params)",
               "rms_norm_static_fp8_quant(RmsNormStaticFp8QuantParams& params)",
               "fused_add_rms_norm_static_fp8_quant(\n    FusedAddRmsNormStaticFp8QuantParams& params)"):
        i = rest.index(fn)
        j = rest.index("#if defined(USE_CUDA)", i)
        assert j - i < 200, fn
        rest = rest[:j] + "#if defined(USE_CUDA)" + MI355 + rest[j + len("#if defined(USE_CUDA)"):]
    return head + sep + rest


def edit_deepseek_v2_attention(src: str) -> str:
    """DeepseekV2AttentionImpl: (1) flash_mla::dense_decode comes from the MI355 shim (same DenseDecodeParams, same call site);
    (2) prefill reads the latent rows back from the paged cache (no host loop over sequences, no .cpu())"""
    a = '#include "kernels/dcu/flash_mla_adapter.h"\n'
    assert a in src
    src = src.replace(a, '#if defined(USE_MI355)\n'
                         '// kernel::dcu::flash_mla::{DenseDecodeParams, dense_decode} of kernels/dcu/flash_mla_adapter.h, served by the MI355X\n'
                         '// backend: decode_flash_mla below compiles unchanged\n'
                         '#include "kernels/mi355/mi355_ops_api.h"\n'
                         'namespace xllm::kernel {\nnamespace dcu = mi355;\n}  // namespace xllm::kernel\n'
                         '#else\n' + a + '#endif\n', 1)
    # (3) the two weight-absorption products leave rocBLAS: kernel::mi355::bmm_heads reads the token-major tensors in place and takes
    #     every head's matrix K-contiguous per output column ([h, N, K]): w_vc_ stays the slice of kv_b_proj's weight (no transpose at
    #     load), w_kc_ is transposed once at load instead
    c = ('  torch::Tensor attn_bmm =\n      torch::bmm(attn_latent.transpose(0, 1), w_vc_);  // [tp_heads, tokens, v]\n'
         '  attn_bmm = attn_bmm.transpose(0, 1);                 // [tokens, tp_heads, v]\n')
    assert c in src
    src = src.replace(c, '#if defined(USE_MI355)\n'
                         '  // w_vc_ is [tp_heads, v, kv_lora] here (see load_state_dict): one launch, no transposes\n'
                         '  torch::Tensor attn_bmm = kernel::mi355::bmm_heads(attn_latent, w_vc_);  // [tokens, tp_heads, v]\n'
                         '#else\n' + c + '#endif\n', 1)
    d = ('  torch::Tensor q_nope_absorbed =\n      torch::bmm(q_nope.transpose(0, 1), w_kc_).transpose(0, 1);\n')
    assert d in src
    src = src.replace(d, '#if defined(USE_MI355)\n'
                         '  // w_kc_ is [tp_heads, kv_lora, qk_nope] here (see load_state_dict)\n'
                         '  torch::Tensor q_nope_absorbed = kernel::mi355::bmm_heads(q_nope, w_kc_);\n'
                         '#else\n' + d + '#endif\n', 1)
    e = ('    w_vc_ = w_vc_.transpose(1, 2)\n                .contiguous();  // [H, v_head, kv_lora] -> [H, kv_lora, v_head]\n')
    assert e in src
    src = src.replace(e, '#if defined(USE_MI355)\n'
                         '    // kernel::mi355::bmm_heads takes [H, N, K]: W_vc as it lies in kv_b_proj ([H, v_head, kv_lora], a view), W_kc\n'
                         '    // transposed once: [H, qk_nope, kv_lora] -> [H, kv_lora, qk_nope]\n'
                         '    w_kc_ = w_kc_.transpose(1, 2).contiguous();\n'
                         '#else\n' + e + '#endif\n', 1)
    b = '  if (is_prefill) {\n    return prefill_sdpa(q_nope_absorbed, q_pe, latent_normed, attn_metadata);\n  }\n'
    assert b in src
    src = src.replace(b, '  if (is_prefill) {\n'
                         '#if defined(USE_MI355)\n'
                         '    // the latent rows of this step are already in the paged cache (store_latent_cache above): one kernel over the\n'
                         '    // block table instead of the per-sequence SDPA loop on the host -- graph-capturable, no device-to-host copy\n'
                         '    if (k_cache.defined() && attn_metadata.slot_mapping.defined() &&\n'
                         '        attn_metadata.block_table.defined()) {\n'
                         '      return project_output(kernel::mi355::flash_mla::prefill_paged(\n'
                         '          q_nope_absorbed, q_pe, k_cache, attn_metadata.q_cu_seq_lens,\n'
                         '          attn_metadata.kv_seq_lens, attn_metadata.block_table, kv_lora_rank_,\n'
                         '          softmax_scale_, attn_metadata.is_causal));\n'
                         '    }\n'
                         '#endif\n'
                         '    return prefill_sdpa(q_nope_absorbed, q_pe, latent_normed, attn_metadata);\n  }\n', 1)
    return src


def edit_qwen2_attention(src: str) -> str:
    # fused_qk_norm_rope is called as xllm::kernel::cuda::fused_qk_norm_rope from the layer
    src = gate_everywhere(src)
    old = '#if defined(USE_CUDA) || defined(USE_DCU) || defined(USE_MI355)\n#include "kernels/cuda/cuda_ops_api.h"\n#endif\n'
    assert old in src
    return src.replace(old, '#if defined(USE_MI355)\n' + MI355_OPS_INCLUDE +
                       '#elif defined(USE_CUDA) || defined(USE_DCU)\n#include "kernels/cuda/cuda_ops_api.h"\n#endif\n', 1)


def edit_worker_impl(src: str) -> str:
    # WorkerImpl::execute_cuda_block_copy_kernel calls xllm::kernel::cuda::block_copy (worker_impl.cpp:1071-1082)
    src = gate_everywhere(src)
    old = ('#elif defined(USE_CUDA) || defined(USE_DCU) || defined(USE_MI355)\n#include "kernels/cuda/cuda_ops_api.h"\n'
           '#include "platform/cuda_profiler.h"\n')
    assert old in src
    return src.replace(old, '#elif defined(USE_MI355)\n' + MI355_OPS_INCLUDE +
                       '#include "platform/cuda_profiler.h"\n#include "platform/torch_profiler.h"\n'
                       '#elif defined(USE_CUDA) || defined(USE_DCU)\n#include "kernels/cuda/cuda_ops_api.h"\n'
                       '#include "platform/cuda_profiler.h"\n', 1)


def edit_setup_py(src: str) -> str:
    old = '        elif self.device == "maca":\n'
    assert old in src
    src = src.replace(old, '''        elif self.device == "mi355":
            # AMD Instinct MI355X (gfx950): the HIP-family host code of the DCU build over libxllm_mi355.so; no
            # flash_attention / flash_mla / aiter libraries. MI355_ROOT = a checkout of the backend.
            import torch

            if not getattr(torch.version, "hip", None):
                raise RuntimeError("MI355 build requires a ROCm PyTorch (torch.version.hip).")
            cmake_args += [
                "-DUSE_MI355=ON",
                f"-DMI355_ROOT={os.environ['MI355_ROOT']}",
                f"-DROCM_PATH={os.getenv('ROCM_PATH', '/opt/rocm')}",
                f"-DTORCH_CMAKE_PREFIX={get_torch_cmake_prefix_path()}",
                f"-DTORCH_PKG_DIR={get_torch_root_path()}",
                "-DCMAKE_HIP_ARCHITECTURES=gfx950",
            ]
            set_dcu_envs()               # the common build environment; the closed DCU libraries it looks for are not used
            os.environ["DCU_PATH"] = os.getenv("ROCM_PATH", "/opt/rocm")
''' + old, 1)
    src = src.replace("choices=['auto', 'npu', 'mlu', 'cuda', 'ilu', 'musa', 'dcu', 'maca'],",
                      "choices=['auto', 'npu', 'mlu', 'cuda', 'ilu', 'musa', 'dcu', 'maca', 'mi355'],", 1)
    return src


def edit_root_cmake(src: str) -> str:
    src = src.replace(
        'option(USE_DCU "Enable DCU support" OFF)\n',
        'option(USE_DCU "Enable DCU support" OFF)\noption(USE_MI355 "Enable AMD Instinct MI355X (gfx950) support" OFF)\n'
        'if(USE_MI355 AND (NOT DEFINED ENV{DCU_PATH} OR "$ENV{DCU_PATH}" STREQUAL ""))\n'
        '  # the HIP-family blocks below (shared with USE_DCU) read the toolkit root from DCU_PATH\n'
        '  set(ENV{DCU_PATH} "/opt/rocm")\nendif()\n', 1)
    src = cmake_gate_everywhere(src)
    old = "if(USE_DCU OR USE_MI355)\n  add_definitions(-DUSE_DCU)\n"
    assert old in src
    return src.replace(
        old,
        "if(USE_DCU OR USE_MI355)\n"
        "  if(USE_MI355)\n"
        "    # MI355_ROOT = a checkout of the backend: include/xllm_mi355.h (C ABI), shim/ (libtorch binding),\n"
        "    # xllm_amd/lib/libxllm_mi355.so (make -C xllm_amd/csrc: hipcc --offload-arch=gfx950)\n"
        "    add_definitions(-DUSE_MI355)\n"
        "    include_directories(${MI355_ROOT}/include)\n"
        "    link_directories(${MI355_ROOT}/xllm_amd/lib)\n"
        "  else()\n"
        "    add_definitions(-DUSE_DCU)\n"
        "  endif()\n", 1)


def edit_kernels_cmake(src: str) -> str:
    src = src.replace(
        "if(USE_DCU)\n  add_subdirectory(dcu)\nendif()\n",
        "if(USE_DCU)\n  add_subdirectory(dcu)\nendif()\n\n"
        "if(USE_MI355)\n"
        "  # kernels/mi355/ = ${MI355_ROOT}/shim/mi355_ops_api.{h,cpp} (a symlink or a copy); the kernels themselves are in\n"
        "  # libxllm_mi355.so behind the C ABI. dcu/{attention_runner,piecewise_graphs,global_capture_instance}.cpp are the\n"
        "  # reference's own HOST code for piecewise HIP-graph capture (ATen HIPGraph only): runtime/dcu_graph_executor_impl.cpp\n"
        "  # and layers/mi355/attention.cpp use them unchanged\n"
        "  cc_library(NAME mi355_kernels\n"
        "    HDRS mi355/mi355_ops_api.h dcu/attention_runner.h dcu/piecewise_graphs.h dcu/global_capture_instance.h\n"
        "    SRCS mi355/mi355_ops_api.cpp dcu/attention_runner.cpp dcu/piecewise_graphs.cpp dcu/global_capture_instance.cpp\n"
        "    DEPS hip::host torch glog::glog :platform xllm_mi355)\n"
        "endif()\n", 1)
    old = "    $<$<BOOL:${USE_DCU}>:dcu_kernels>\n"
    assert old in src
    return src.replace(old, old + "    $<$<BOOL:${USE_MI355}>:mi355_kernels>\n", 1)


def edit_layers_cmake(src: str) -> str:
    old = "elseif(USE_DCU)\n  add_subdirectory(dcu)\n"
    assert old in src
    return src.replace(
        old, old + "elseif(USE_MI355)\n"
        "  # layers/mi355/ = ${MI355_ROOT}/shim/layers_mi355/CMakeLists.txt + ${MI355_ROOT}/shim/mi355_attention.{h,cpp} as\n"
        "  # attention.{h,cpp}; the target also compiles ../dcu/{fused_moe,deepseek_v2_attention,deepseek_v2_decoder_layer_impl}.cpp\n"
        "  add_subdirectory(mi355)\n", 1)


def edit_layers_common_cmake(src: str) -> str:
    # the routed-expert layer and the MLA attention / decoder layer of the DCU backend (layers/dcu/{fused_moe,deepseek_v2_attention,
    # deepseek_v2_decoder_layer_impl}.cpp) are host code over kernel::* operators and flash_mla::dense_decode: a USE_MI355 build
    # compiles THOSE files (layers/mi355/CMakeLists.txt lists them) and keeps layers/common/fused_moe.* out, exactly as USE_DCU does
    n = src.count("$<BOOL:${USE_CUDA}>,$<BOOL:${USE_DCU}>>>:fused_moe.")
    assert n == 2
    src = src.replace("$<BOOL:${USE_CUDA}>,$<BOOL:${USE_DCU}>>>:fused_moe.", "$<BOOL:${USE_CUDA}>,$<BOOL:${USE_DCU}>,$<BOOL:${USE_MI355}>>>:fused_moe.")
    old = "    $<$<BOOL:${USE_DCU}>:dcu_layers>\n"
    assert old in src
    return src.replace(old, old + "    $<$<BOOL:${USE_MI355}>:mi355_layers>\n", 1)


def edit_attention_h(src: str) -> str:
    old = '#elif defined(USE_DCU)\n#include "layers/dcu/attention.h"\n'
    assert old in src
    return src.replace(old, old + '#elif defined(USE_MI355)\n#include "layers/mi355/attention.h"\n', 1)


SPECIFIC = {
    "CMakeLists.txt": edit_root_cmake,
    "setup.py": edit_setup_py,
    "xllm/core/kernels/CMakeLists.txt": edit_kernels_cmake,
    "xllm/core/layers/CMakeLists.txt": edit_layers_cmake,
    "xllm/core/layers/common/CMakeLists.txt": edit_layers_common_cmake,
    "xllm/core/layers/common/attention.h": edit_attention_h,
    "xllm/core/layers/common/qwen2_attention.cpp": edit_qwen2_attention,
    "xllm/core/layers/dcu/deepseek_v2_attention.cpp": edit_deepseek_v2_attention,     # no USE_DCU inside: listed by hand
    "xllm/core/kernels/ops_api.cpp": edit_ops_api,
    "xllm/core/runtime/worker_impl.cpp": edit_worker_impl,
}

# what the walk test accepts as the difference between `original under USE_DCU` and `patched under USE_MI355`:
# file -> (lines only the USE_DCU build has, lines only the USE_MI355 build has), whitespace-normalised
_ALIAS_CUDA = ['#include "kernels/mi355/mi355_ops_api.h"', "namespace xllm::kernel {", "namespace cuda = mi355;", "} // namespace xllm::kernel"]
EXPECTED_DIFF = {
    "xllm/core/kernels/ops_api.cpp": (
        ['#include "cuda/cuda_ops_api.h"', '#include "dcu/dcu_ops_api.h"'],
        ["// AMD Instinct MI355X (gfx950): xllm::kernel::mi355::* has the argument lists of the cuda:: / dcu:: functions the DCU",
         "// branches below call, so those branches are reused verbatim through two namespace aliases",
         '#include "mi355/mi355_ops_api.h"', "namespace xllm::kernel {", "namespace cuda = mi355;", "namespace dcu = mi355;",
         "} // namespace xllm::kernel", "FP8_BRANCHES"]),          # + the five fp8 operators (CUDA-only in the reference)
    "xllm/core/layers/common/attention.h": (['#include "layers/dcu/attention.h"'], ['#include "layers/mi355/attention.h"']),
    "xllm/core/layers/common/qwen2_attention.cpp": (['#include "kernels/cuda/cuda_ops_api.h"'], _ALIAS_CUDA),
    "xllm/core/runtime/worker_impl.cpp": (['#include "kernels/cuda/cuda_ops_api.h"'], _ALIAS_CUDA),
}


def discover():
    """every non-test file of the reference that mentions USE_DCU -> 'H-cpp' | 'H-cmake' | 'K' | 'replaced'"""
    found = {}
    for d, _, fs in os.walk(REF):
        rel_d = os.path.relpath(d, REF)
        if rel_d.split(os.sep)[0] in ("tests", ".git", "third_party", "docs"):
            continue
        for f in fs:
            p = os.path.join(d, f)
            rel = os.path.normpath(os.path.join(rel_d, f))
            if not (f.endswith((".cpp", ".h", ".cu", ".cuh", ".hip", ".py")) or f == "CMakeLists.txt"):
                continue
            try:
                if "USE_DCU" not in open(p, errors="replace").read():
                    continue
            except OSError:
                continue
            if rel.startswith(REPLACED_PREFIXES):
                found[rel] = "replaced"
            elif rel in SPECIFIC:
                found[rel] = "K"
            elif f == "CMakeLists.txt":
                found[rel] = "H-cmake"
            else:
                found[rel] = "H-cpp"
    return dict(sorted(found.items()))


def edits():
    plan = {}
    for rel, kind in discover().items():
        if kind == "H-cpp":
            plan[rel] = gate_everywhere
        elif kind == "H-cmake":
            plan[rel] = cmake_gate_everywhere
    plan.update(SPECIFIC)
    return dict(sorted(plan.items()))


def gate_table():
    """markdown rows: file | gate lines (reference numbering) | what the DCU / MI355 branch holds"""
    rows = []
    for rel, kind in discover().items():
        src = open(os.path.join(REF, rel), errors="replace").read().split("\n")
        at = [str(i + 1) for i, l in enumerate(src) if "USE_DCU" in l]
        rule = {"H-cpp": "H", "H-cmake": "H (CMake)", "K": "K (specific edit)", "replaced": "not compiled under USE_MI355"}[kind]
        rows.append(f"| `{rel}` | {', '.join(at)} | {rule} |")
    return rows


def main():
    global REF
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    if pos:
        REF = pos[0]
    if "--gates" in sys.argv:
        print("\n".join(gate_table()))
        return
    chunks, nfiles = [], 0
    for rel, fn in edits().items():
        old = open(os.path.join(REF, rel)).read()
        new = fn(old)
        assert new != old, f"no edit applied to {rel}"
        nfiles += 1
        chunks += difflib.unified_diff(old.splitlines(True), new.splitlines(True), "a/" + rel, "b/" + rel, n=3)
    header = ("# xllm-use-mi355.patch -- generated by tools/make_reference_patch.py against the reference tree; apply with\n"
              "#   git apply xllm-use-mi355.patch   (then: ln -s $MI355_ROOT/shim xllm/core/kernels/mi355, and\n"
              "#   xllm/core/layers/mi355/{attention.h,attention.cpp} from $MI355_ROOT/shim/mi355_attention.*, CMakeLists.txt from\n"
              "#   $MI355_ROOT/shim/layers_mi355/ -- it also compiles layers/dcu/{fused_moe,deepseek_v2_attention,deepseek_v2_decoder_layer_impl}.cpp;\n"
              "#   build with `python setup.py bdist_wheel --device mi355` or -DUSE_MI355=ON -DMI355_ROOT=...). See INTEGRATION.md section 3.\n")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(header + "".join(chunks))
    n = sum(1 for c in chunks if c.startswith("@@"))
    print(f"wrote {OUT}: {nfiles} files, {n} hunks")


if __name__ == "__main__":
    main()
