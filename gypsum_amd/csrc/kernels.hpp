// kernels.hpp -- the __global__ kernels of libgypsum_hip.so (gfx950 only).
//
//   corr_cells_kernel   one workgroup per (stream, satellite, Doppler) cell, min(K, 8) wavefronts (K = samples per
//                       chip; K > 8 in rounds of 8 polyphase branches): per ms block carrier wipe-off + polyphase
//                       pre-sum (global -> LDS), then per wavefront FFT2048 -> x conj(PRN spectrum) -> IFFT2048 and
//                       |.| accumulation in registers (non-coherent), or the blocks folded before ONE transform
//                       (coherent); finally max / first-argmax / sum / count-of-max reductions.
//                       == utils.py:77-108 + the reductions of acquisition.py:180-189.
//   corr_cells_pipe_kernel  the same non-coherent cell for K == 8 (8.184 Msps, the acquisition search of the headline
//                       workload), software-pipelined: one workgroup per CU, 256 VGPRs, double-buffered rows in LDS,
//                       next block's samples prefetched during the transforms, replica spectrum resident in registers.
//   grid_fold_kernel / grid_cells[_wave]_kernel   flat grids whose satellites share the Doppler bins: wipe-off and
//                       pre-sum once per (stream, bin), then transform-only workgroups / wavefronts per satellite.
//   track_step_kernel   same core for one explicit millisecond of one tracking channel, plus the early/late taps
//                       and the rolled-PRN argmax of tracker.py:284-313.
//   track_block_kernel  persistent per-channel loop over many milliseconds with the DLL / Costas / lock-detector
//                       / circularity-watchdog state on the device (tracker.py:157-203, 246-262, 297-305, 331-389).
//                       MODE 0: throughput form.  MODE 2 (K = 8, 2; banks of at most one channel per CU): speculative
//                       form -- window correlations around the last peak lag on the serial chain, only the lock verdict
//                       between a peak and the next wipe-off, track_verify_kernel proving every such millisecond's
//                       arg-max from the full profile in parallel.
//   acq_* kernels       the level-to-level bookkeeping of acquisition.py:70-152 on the device: plan, work list, record
//                       reuse (optional), float64 tie-breaks within a level (refine) and across levels (exact).
//
// The kernels live in the kernels_*.hpp parts below, one per family.
#pragma once
#include "kernels_common.hpp"
#include "kernels_cells.hpp"
#include "kernels_grid.hpp"
#include "kernels_track_step.hpp"
#include "kernels_track_block.hpp"
#include "kernels_dll_exact.hpp"
#include "kernels_acq.hpp"
#include "kernels_misc.hpp"
