"""The optional kernel families of the training iteration: which exist, which GPU tests hold each to the framework operators,
which are ON by default (the COMMITTED list: what ``bench.py`` measures and what ``tools/train_val.py`` trains with), and the
function that switches them at run time.  A family is listed in COMMITTED only with green tests
(``tests/test_kernel_families_cpu.py`` checks that the tests named here exist).

Each family can also be forced from the environment (``MDETR_<FAMILY>=1``) -- for A/B runs: any such variable replaces the
committed list altogether (``MDETR_BENCH_DEFAULT_PATH=1`` / ``trainer.kernels: default`` = none at all)."""
import os

# Each can be forced with an environment variable of the same name (= "1").  Without any in the environment the step runs
# with the COMMITTED list below.  (MDETR_MSDA_BF16 changes the MSDA operator's element types; the roofline accounting
# follows it: msda_algorithmic_bytes(mixed=True).)
AUTOTUNE_SWITCHES = ("MDETR_FUSED_LOSSES", "MDETR_FUSED_ADAMW", "MDETR_MSDA_PROLOGUE", "MDETR_FUSED_LN", "MDETR_MSDA_BF16",
                     "MDETR_FUSED_EPILOGUE", "MDETR_GEMM_RELU", "MDETR_CONV3X3", "MDETR_GROUP_NORM", "MDETR_SMALL_WGRAD",
                     "MDETR_CONV_STRIDED", "MDETR_CONV_WGRAD", "MDETR_CONV_STEM", "MDETR_TGEMM", "MDETR_WFOLD", "MDETR_RELU_PREMASK",
                     "MDETR_HEADS", "MDETR_CHUNK_SUMS", "MDETR_HEAD_TAIL")
ALL_SWITCHES = AUTOTUNE_SWITCHES
# The measured configuration.  family -> the GPU tests that hold it to the default path / the framework operators
# (all in tests/test_fused_gpu.py unless a file is named); a family without green tests is not listed.
SWITCH_TESTS = {
    "MDETR_FUSED_LOSSES": "test_fused_pair_losses_*, test_fused_ddn_loss_*, test_fused_cost_solver_*, test_training_step_with_fused_criterion_*",
    "MDETR_FUSED_ADAMW": "test_fused_adamw_kernel_*, test_fused_adamw_vs_recorded_reference_steps, test_training_step_with_fused_adamw_*",
    "MDETR_FUSED_LN": "test_fused_add_layernorm_*, test_training_step_with_fused_layernorm_*",
    "MDETR_MSDA_PROLOGUE": "test_msda_prologue_kernel_*",
    "MDETR_MSDA_BF16": "test_msda_bf16_kernels_*, test_msda_function_with_native_bf16_*, test_training_step_with_bf16_msda_*, test_msda_gpu.py::test_bf16_native_full_encoder_shape_vs_oracle",
    "MDETR_FUSED_EPILOGUE": "test_bias_act_kernel_*, test_training_step_with_fused_tails_*",
    "MDETR_GEMM_RELU": "test_library_gemm_relu_epilogue_*, test_training_step_with_fused_tails_*",
    "MDETR_SMALL_WGRAD": "test_small_wgrad_kernel_*, test_training_step_with_the_small_wgrad_kernel_*",
    "MDETR_GROUP_NORM": "test_group_norm_kernel_*, test_training_step_with_the_group_norm_kernel_*",
    "MDETR_CONV_STRIDED": "test_conv_strided_kernel_matches_the_library_convolution, test_training_step_with_the_convolution_kernels_*",
    "MDETR_CONV_WGRAD": "test_conv_wgrad_kernel_matches_the_library_weight_gradient, test_conv_strided_kernel_*, test_training_step_with_the_convolution_kernels_*",
    "MDETR_CONV_STEM": "test_conv_stem_kernel_matches_the_library_convolution, test_training_step_with_the_convolution_kernels_*",
    "MDETR_TGEMM": "test_tgemm_gpu.py::test_tgemm_*, test_training_step_with_the_token_gemm_kernel_*, test_bottleneck_with_fused_tails_*",
    "MDETR_WFOLD": "test_fold_kernel_*, test_training_step_with_the_fold_kernel_*",
    "MDETR_RELU_PREMASK": "test_tgemm_gpu.py::test_masked_input_gradient_*, test_tgemm_gpu.py::test_bottleneck_stage_with_premasked_relu_*",
    "MDETR_HEADS": "test_sgemm_gpu.py::test_sgemm_*, test_sgemm_gpu.py::test_heads_level_*, test_training_step_with_the_grouped_heads_*",
    "MDETR_HEAD_TAIL": "test_sgemm_gpu.py::test_head_tail_*, test_sgemm_gpu.py::test_training_step_with_the_head_tail_*",
    "MDETR_CHUNK_SUMS": "test_colsum_gpu.py::test_chunk_sums_*, test_colsum_gpu.py::test_deferred_chunk_sums_*",
    "MDETR_CONV3X3": "test_conv3x3_kernel_matches_the_library_convolution, test_training_step_with_the_conv3x3_kernel_*, test_conv3x3_module_with_a_trainable_bias_*",
}
COMMITTED_SWITCHES = {
    # (MDETR_CONV3X3: 1.6-3.5x MIOpen per kernel on the four ResNet stages, profiles/r02a_fusedbench.json; the step 249.3 vs
    # 234.6 img/s, profiles/r02b_bench_committed_plus_conv3x3.json.
    # MDETR_GROUP_NORM: 328.9 vs 308.7 img/s under graph replay, profiles/r02m_bench_with_gn.json.
    # MDETR_SMALL_WGRAD: 340.2 vs 333.6 img/s, profiles/r02q_bench_{with_small_wgrad,committed}.json.)
    # MDETR_CONV_WGRAD / _STRIDED / _STEM (round 3): every convolution of the backbone, the pyramid and the depth head by hand --
    # weight gradients 1.1-1.6x MIOpen's, the stem 2.9x, the stride-2 forms 0.4-2.8x (profiles/r03c_convbench.json); the step
    # 354.2 vs 348.0 img/s (r03d_bench_{all,committed}.json), no MIOpen kernel left in the iteration.)
    # MDETR_TGEMM (round 5): every token-wise product of the bf16 step (1x1 convolutions, linear layers; forward and input gradient)
    # through csrc/tgemm.hip with the tails in its epilogue: 396.6 -> 418.5 img/s in one call (profiles/r05f_step_ab.log).
    # MDETR_WFOLD (round 5): the frozen-BN fold of the 42 trainable backbone weights, their [C][tap][O] copies and the unfolding of
    # their gradients as one launch each way (csrc/wfold.hip): 426.3 -> 432.4 img/s in one call (profiles/r05z1_step_ab_wfold.log).
    # MDETR_RELU_PREMASK (round 5): the ReLU backward masks inside the bottlenecks (conv2 -> conv3) and between consecutive blocks of a
    # stage applied in the consumers' input-gradient products (mdetr_tgemm_masked), 23 elementwise passes less: 436.2 -> 441.4 img/s in
    # one call, bit-identical gradients (profiles/r05z2_step_ab_premask.log).
    "bf16": ("MDETR_FUSED_LOSSES", "MDETR_FUSED_ADAMW", "MDETR_FUSED_LN", "MDETR_MSDA_PROLOGUE", "MDETR_MSDA_BF16",
             "MDETR_FUSED_EPILOGUE", "MDETR_GEMM_RELU", "MDETR_CONV3X3", "MDETR_GROUP_NORM", "MDETR_SMALL_WGRAD",
             "MDETR_CONV_WGRAD", "MDETR_CONV_STRIDED", "MDETR_CONV_STEM", "MDETR_TGEMM", "MDETR_WFOLD", "MDETR_RELU_PREMASK", "MDETR_HEADS",
             "MDETR_CHUNK_SUMS", "MDETR_HEAD_TAIL"),
    "fp32": ("MDETR_FUSED_LOSSES", "MDETR_FUSED_ADAMW", "MDETR_FUSED_LN", "MDETR_MSDA_PROLOGUE",
             "MDETR_FUSED_EPILOGUE", "MDETR_GEMM_RELU", "MDETR_GROUP_NORM", "MDETR_SMALL_WGRAD", "MDETR_HEADS",
             "MDETR_HEAD_TAIL"),
}
COMMITTED_SWITCHES["bf16-autocast"] = COMMITTED_SWITCHES["fp32"]


def committed_switches(precision):
    """(switch set, source): the environment's MDETR_* if any is set (A/B experiments), else the committed list."""
    env = env_switches()
    if env or os.environ.get("MDETR_BENCH_DEFAULT_PATH") == "1":
        return env, "environment"
    return set(COMMITTED_SWITCHES[precision]), "bench.COMMITTED_SWITCHES"


def env_switches():
    return {k for k in ALL_SWITCHES if os.environ.get(k) == "1"}


def apply_switches(names):
    """Runtime equivalent of the environment switches for the module-level ones (the criterion's and the optimizer's
    are applied by TrainStep)."""
    from monodetr_amd import chunk_sums, head_tail_ext
    chunk_sums.ENABLED = "MDETR_CHUNK_SUMS" in names
    head_tail_ext.ENABLED = "MDETR_HEAD_TAIL" in names
    from monodetr_amd import add_ln_ext, bias_act_ext, conv3x3_ext, conv_stem_ext, conv_taps_ext, conv_wgrad_ext, group_norm_ext, small_wgrad_ext, wfold_ext
    from monodetr_amd.monodetr import heads, linear
    from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func
    from monodetr_amd.monodetr.ops.modules import ms_deform_attn
    ms_deform_attn._FUSED_PROLOGUE = "MDETR_MSDA_PROLOGUE" in names
    add_ln_ext.ENABLED = "MDETR_FUSED_LN" in names
    linear._TGEMM = "MDETR_TGEMM" in names
    heads.ENABLED = "MDETR_HEADS" in names
    wfold_ext.ENABLED = "MDETR_WFOLD" in names
    linear._PREMASK = "MDETR_RELU_PREMASK" in names
    linear._GEMM_RELU = "MDETR_GEMM_RELU" in names
    bias_act_ext.ENABLED = "MDETR_FUSED_EPILOGUE" in names
    conv3x3_ext.ENABLED = "MDETR_CONV3X3" in names
    conv_taps_ext.ENABLED = "MDETR_CONV_STRIDED" in names
    conv_wgrad_ext.ENABLED = "MDETR_CONV_WGRAD" in names
    conv_stem_ext.ENABLED = "MDETR_CONV_STEM" in names
    group_norm_ext.ENABLED = "MDETR_GROUP_NORM" in names
    small_wgrad_ext.ENABLED = "MDETR_SMALL_WGRAD" in names
    ms_deform_attn_func._NATIVE_BF16 = "MDETR_MSDA_BF16" in names


def enable_for_training(model, criterion, optimizer_cfg, precision="fp32", choice="committed"):
    """What ``tools/train_val.py`` does before it builds the optimizer: switch on the committed families for `precision`
    (or the environment's, or none for choice == "default"), set the criterion's flags, and return the optimizer config
    with the fused AdamW selected accordingly.  Returns (optimizer_cfg, sorted names)."""
    names, _ = (set(), "default") if choice == "default" else committed_switches("bf16" if precision == "bf16" else "fp32")
    apply_switches(names)
    criterion.fused_pair_losses = criterion.matcher.fused_cost = "MDETR_FUSED_LOSSES" in names
    return dict(optimizer_cfg, fused="MDETR_FUSED_ADAMW" in names or bool(optimizer_cfg.get("fused", False))), sorted(names)
