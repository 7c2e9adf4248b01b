"""THE definition of "the MFMA family" shared by bench.py (FLOP hooks, per-kernel time / MFMA-busy tables) and the PMC
tools (tools/pmc_traffic.py): the kernels that launch through IVLN_LAUNCH_FAMILY (csrc/family_timing.h).  A mirror of
`ivln_family_kernel_names()` in libivln_hip.so - tests/test_cabi_exports.py pins the mirror to the library and to the
launch sites in csrc/*.hip, so a new kernel form cannot drop out of `roofline.traffic` again (the round's new
k_conv1x1_bf3_ks was missed by a prefix list)."""
import re

FAMILY_KERNELS = ("k_gemm", "k_gemm_vec", "k_conv_direct", "k_wgrad_direct", "k_conv1x1_stream", "k_conv_bf3",
                  "k_conv_bf3_ks", "k_conv1x1_bf3_ks", "k_conv7s2_bf3", "k_wgrad_bf3", "k_gn_conv", "k_nconv", "k_depth_net")
MAPPER_KERNELS = ("k_local_minmax", "k_local_argmax", "k_local_select", "k_world_max", "k_world_select", "k_finalize",
                  "k_frames", "k_swap_counts")


def base_name(kernel):
    """'void (anonymous namespace)::k_conv_bf3_ks<4, 16, 2>(desc, ...)' -> 'k_conv_bf3_ks'."""
    k = re.sub(r"\(anonymous namespace\)::", "", kernel.strip().strip('"'))
    k = re.sub(r"^void\s+", "", k)
    return re.split(r"[<(\s]", k, maxsplit=1)[0]


def is_family(kernel):
    return base_name(kernel) in FAMILY_KERNELS


def is_mapper(kernel):
    return base_name(kernel) in MAPPER_KERNELS
