#!/usr/bin/env python3
"""Writes patches/xllm-use-mi355.patch: the reference-side binding of the MI355X backend, every hunk written out.

The reference dispatches its operators at compile time (`#if defined(USE_<BACKEND>)` chains in
xllm/core/kernels/ops_api.cpp) and its DCU branches call `cuda::<op>` / `dcu::<op>` with exactly the argument lists
shim/mi355_ops_api.h declares under `xllm::kernel::mi355`. So the binding is: (1) include the MI355 header and alias the two
namespaces to it, (2) let every DCU branch (and the CUDA-only fp8 branches) also compile under USE_MI355, (3) pick
layers/mi355/attention.h, (4) build flags. This script applies those edits to a scratch copy of the reference files and diffs
them -- run it where /root/reference exists; the patch it writes is committed (context lines are the reference's, as in any
patch; no reference source is copied into this repository otherwise).
"""
import difflib
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "patches", "xllm-use-mi355.patch")
MI355 = " || defined(USE_MI355)"


def add_to_dcu_conditions(src: str) -> str:
    """every preprocessor condition that mentions USE_DCU also accepts USE_MI355"""
    def fix(m):
        line = m.group(0)
        return line if "USE_MI355" in line else line.replace("defined(USE_DCU)", "defined(USE_DCU)" + MI355)
    return re.sub(r"^[ \t]*#[ \t]*(?:if|elif)[^\n]*defined\(USE_DCU\)[^\n]*$", fix, src, flags=re.M)


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
    rest = add_to_dcu_conditions(rest)
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


EDITS = {
    "CMakeLists.txt": lambda s: s.replace(
        'option(USE_DCU "Enable DCU support" OFF)\n',
        'option(USE_DCU "Enable DCU support" OFF)\noption(USE_MI355 "Enable AMD Instinct MI355X (gfx950) support" OFF)\n', 1).replace(
        "if(USE_DCU)\n  add_definitions(-DUSE_DCU)\n",
        "if(USE_MI355)\n"
        "  # MI355_ROOT = a checkout of the backend: include/xllm_mi355.h (C ABI), shim/ (libtorch binding),\n"
        "  # xllm_amd/lib/libxllm_mi355.so (make -C xllm_amd/csrc: hipcc --offload-arch=gfx950)\n"
        "  add_definitions(-DUSE_MI355)\n"
        "  include_directories($ENV{PYTHON_INCLUDE_PATH})\n"
        "  include_directories(SYSTEM $ENV{PYTORCH_INSTALL_PATH}/include\n"
        "                      $ENV{PYTORCH_INSTALL_PATH}/include/torch/csrc/api/include)\n"
        "  include_directories(${MI355_ROOT}/include)\n"
        "  link_directories($ENV{PYTORCH_INSTALL_PATH}/lib ${MI355_ROOT}/xllm_amd/lib)\n"
        "endif()\n\n"
        "if(USE_DCU)\n  add_definitions(-DUSE_DCU)\n", 1),
    "xllm/core/kernels/CMakeLists.txt": lambda s: s.replace(
        "if(USE_DCU)\n  add_subdirectory(dcu)\nendif()\n",
        "if(USE_DCU)\n  add_subdirectory(dcu)\nendif()\n\n"
        "if(USE_MI355)\n"
        "  # kernels/mi355/ = ${MI355_ROOT}/shim/mi355_ops_api.{h,cpp} (a symlink or a copy); the kernels themselves are in\n"
        "  # libxllm_mi355.so behind the C ABI\n"
        "  cc_library(NAME mi355_kernels HDRS mi355/mi355_ops_api.h SRCS mi355/mi355_ops_api.cpp DEPS torch xllm_mi355)\n"
        "endif()\n", 1).replace(
        "    $<$<BOOL:${USE_DCU}>:dcu_kernels>\n",
        "    $<$<BOOL:${USE_DCU}>:dcu_kernels>\n    $<$<BOOL:${USE_MI355}>:mi355_kernels>\n", 1),
    "xllm/core/layers/CMakeLists.txt": lambda s: s.replace(
        "elseif(USE_DCU)\n  add_subdirectory(dcu)\n",
        "elseif(USE_DCU)\n  add_subdirectory(dcu)\nelseif(USE_MI355)\n"
        "  # layers/mi355/ = ${MI355_ROOT}/shim/layers_mi355/CMakeLists.txt + ${MI355_ROOT}/shim/mi355_attention.{h,cpp} as\n"
        "  # attention.{h,cpp}; the target also compiles ../dcu/{fused_moe,deepseek_v2_attention,deepseek_v2_decoder_layer_impl}.cpp\n"
        "  add_subdirectory(mi355)\n", 1),
    # the routed-expert layer and the MLA attention / decoder layer of the DCU backend (layers/dcu/{fused_moe,deepseek_v2_attention,
    # deepseek_v2_decoder_layer_impl}.cpp) are host code over kernel::* operators and flash_mla::dense_decode: a USE_MI355 build
    # compiles THOSE files (layers/mi355/CMakeLists.txt lists them) and keeps layers/common/fused_moe.* out, exactly as USE_DCU does
    "xllm/core/layers/common/CMakeLists.txt": lambda s: s.replace(
        "$<BOOL:${USE_CUDA}>,$<BOOL:${USE_DCU}>>>:fused_moe.h>", "$<BOOL:${USE_CUDA}>,$<BOOL:${USE_DCU}>,$<BOOL:${USE_MI355}>>>:fused_moe.h>", 1).replace(
        "$<BOOL:${USE_CUDA}>,$<BOOL:${USE_DCU}>>>:fused_moe.cpp>", "$<BOOL:${USE_CUDA}>,$<BOOL:${USE_DCU}>,$<BOOL:${USE_MI355}>>>:fused_moe.cpp>", 1).replace(
        "    $<$<BOOL:${USE_DCU}>:dcu_layers>\n", "    $<$<BOOL:${USE_DCU}>:dcu_layers>\n    $<$<BOOL:${USE_MI355}>:mi355_layers>\n", 1),
    "xllm/core/layers/qwen3_moe_decoder_layer.h": add_to_dcu_conditions,      # -> layers/dcu/fused_moe.h (FusedMoEImpl)
    "xllm/models/llm/deepseek_v2.h": add_to_dcu_conditions,                   # -> layers/dcu/deepseek_v2_decoder_layer_impl.h
    "xllm/models/models.h": add_to_dcu_conditions,                            # the model list of the DCU build (deepseek_v2, qwen3_moe, ...)
    "xllm/core/layers/dcu/deepseek_v2_attention.cpp": edit_deepseek_v2_attention,
    "xllm/core/kernels/ops_api.cpp": edit_ops_api,
    "xllm/core/layers/common/attention.h": lambda s: s.replace(
        '#elif defined(USE_DCU)\n#include "layers/dcu/attention.h"\n',
        '#elif defined(USE_DCU)\n#include "layers/dcu/attention.h"\n#elif defined(USE_MI355)\n#include "layers/mi355/attention.h"\n', 1),
    # fused_qk_norm_rope is called as xllm::kernel::cuda::fused_qk_norm_rope from the layer
    "xllm/core/layers/common/qwen2_attention.cpp": lambda s: add_to_dcu_conditions(s).replace(
        '#if defined(USE_CUDA) || defined(USE_DCU) || defined(USE_MI355)\n#include "kernels/cuda/cuda_ops_api.h"\n#endif\n',
        '#if defined(USE_MI355)\n#include "kernels/mi355/mi355_ops_api.h"\nnamespace xllm::kernel {\nnamespace cuda = mi355;\n}\n'
        '#elif defined(USE_CUDA) || defined(USE_DCU)\n#include "kernels/cuda/cuda_ops_api.h"\n#endif\n', 1),
    # dcu_w8a8_dynamic_linear_forward (scaled_quantize -> scaled_matmul through ops_api) is backend-neutral
    "xllm/core/layers/common/linear.cpp": add_to_dcu_conditions,
    # c10d::ProcessGroupNCCL is RCCL on a ROCm build of torch: the CUDA / DCU process group is used as is
    "xllm/core/framework/parallel_state/process_group.cpp": add_to_dcu_conditions,
    "xllm/core/framework/parallel_state/collective_communicator.cpp": add_to_dcu_conditions,
    "xllm/core/framework/parallel_state/parallel_state.cpp": lambda s: s.replace(
        "    defined(USE_DCU)\n", "    defined(USE_DCU) || defined(USE_MI355)\n", 1),
}


def main():
    chunks = []
    for rel, fn in EDITS.items():
        old = open(os.path.join(REF, rel)).read()
        new = fn(old)
        assert new != old, f"no edit applied to {rel}"
        chunks += difflib.unified_diff(old.splitlines(True), new.splitlines(True), "a/" + rel, "b/" + rel, n=3)
    header = ("# xllm-use-mi355.patch -- generated by tools/make_reference_patch.py against the reference tree; apply with\n"
              "#   git apply xllm-use-mi355.patch   (then: ln -s $MI355_ROOT/shim xllm/core/kernels/mi355, and\n"
              "#   xllm/core/layers/mi355/{attention.h,attention.cpp} from $MI355_ROOT/shim/mi355_attention.*, CMakeLists.txt from\n"
              "#   $MI355_ROOT/shim/layers_mi355/ -- it also compiles layers/dcu/{fused_moe,deepseek_v2_attention,deepseek_v2_decoder_layer_impl}.cpp;\n"
              "#   configure with -DUSE_MI355=ON -DMI355_ROOT=...). See INTEGRATION.md section 3.\n")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(header + "".join(chunks))
    n = sum(1 for c in chunks if c.startswith("@@"))
    print(f"wrote {OUT}: {len(EDITS)} files, {n} hunks")


if __name__ == "__main__":
    main()
