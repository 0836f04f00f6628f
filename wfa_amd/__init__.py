"""wfa_amd -- MI355X (gfx950) wavefront alignment behind the shenwei356/wfa Aligner API.

The package holds only what the hot path needs: csrc/ (HIP kernels + the C-ABI of include/wfa_hip.h),
lib/ (the built libwfahip.so) and aligner.py (host-side mirror of the reference's Go API over ctypes).
"""
from .aligner import (  # noqa: F401
    AdaptiveReductionOption, Aligner, AlignmentResult, MultiAligner, pack_pairs, generate_pairs_device, BatchResult, DefaultAdaptiveOption, DefaultOptions,
    DefaultPenalties, ErrEmptySeq, ErrSeqTooLong, MaskLower32, MaxSeqLen, New, Op, OpD, OpH, OpI, OpM, OpX,
    Options, Penalties, RecycleAligner, RecycleAlignmentResult, RecycleAlignmentText, WfaError, generate_pairs,
    make_blob, plot_component, trimOps,
)
