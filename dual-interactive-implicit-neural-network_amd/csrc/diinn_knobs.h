// diinn_knobs.h -- the diagnostic overrides of the launch functions, in ONE place.
//
// Every knob is read from the environment once, the first time any launch function asks (a thread-safe static
// initialisation: no getenv on the launch path afterwards), and can be changed in-process through the test-only
// C ABI pair diinn_debug_set / diinn_debug_get (include/diinn_hip.h).  0 / -1 = "not forced" as listed; none of
// them changes results beyond the documented equivalences of the kernel variants (tests/test_gpu_parity.py).
#pragma once
#include <atomic>

struct DiinnKnobs {
    std::atomic<long long> f32_kernel;          // DIINN_F32_KERNEL: 1 throughput, 3 16-pixel latency decode kernel (0: the cheaper by the launch cost model; 2 was round 2's 32-pixel form, deleted)
    std::atomic<long long> bf16_kernel;         // DIINN_BF16_KERNEL: 1 / 2 tiles per wave, 8 cooperative waves (one block per workgroup), 9 the same with persistent workgroups (0: auto)
    std::atomic<long long> x3_kernel;           // DIINN_X3_KERNEL: 1 one block per workgroup, 2 persistent split-bf16 decode (0: auto)
    std::atomic<long long> pbf16_kernel;        // DIINN_PBF16_KERNEL: 1 narrow, 2 wide bf16 P kernel (0: auto)
    std::atomic<long long> p_kernel;            // DIINN_P_KERNEL: 1 direct, 2 Winograd fp32 P kernel (0: auto)
    std::atomic<long long> p_x3_min;            // DIINN_P_X3_MIN: cells from which DIINN_COMPUTE_BF16X3 runs the split-bf16 P kernel (default 32768)
    std::atomic<long long> p_wino_min;          // DIINN_P_WINO_MIN: cells from which the Winograd P kernel runs (default 0)
    std::atomic<long long> enc_s1_min_blocks;   // DIINN_ENC_S1_MIN_BLOCKS (default 128)
    std::atomic<long long> enc_no_stream1x1;    // DIINN_ENC_NO_STREAM1X1 (default 0)
    std::atomic<long long> enc_lat_max_tiles;   // DIINN_ENC_LAT_MAX_TILES (default 256)
    std::atomic<long long> enc_wino_min;        // DIINN_ENC_WINO_MIN (default 8192 pixels)
    std::atomic<long long> enc_wino4_min;       // DIINN_ENC_WINO4_MIN (default -1: by the round count of the two Winograd kernels): n >= 0 = F(4x4,3x3) 3x3 layers from n pixels on
    std::atomic<long long> enc_x3_min;          // DIINN_ENC_X3_MIN (default 32768 pixels): split-bf16 3x3 layers from that map size on
    std::atomic<long long> enc_x3_rows;         // DIINN_ENC_X3_ROWS: split-bf16 3x3 kernel form: 1 / 2 pixel rows per wave, 3 = 2 rows + a group's weights in registers, 4 = that with eight waves (0: auto)
    std::atomic<long long> enc_wino_half_max;   // DIINN_ENC_WINO_HALF_MAX (default -1: by the busiest CU's load)
    std::atomic<long long> enc_wino_persist;    // DIINN_ENC_WINO_PERSIST (default 256 blocks)
    std::atomic<long long> debug_ncu;           // DIINN_DEBUG_NCU (TEST ONLY; default 0: ask the device): the compute-unit count the launch cost models, the F(4x4) rounds / split plan and the persistent grids assume (any value is correct, only differently fast)
    std::atomic<long long> train_split_head;    // DIINN_TRAIN_SPLIT_HEAD (TEST / A-B ONLY; default 0): 1 = the backward pass runs bwd_head_kernel as its own launch instead of inside layer 3's bwd_layer_kernel (bit-identical planes)
    std::atomic<long long> enc_wino4_fault;     // DIINN_ENC_WINO4_FAULT (TEST ONLY; default 0): 1 = split parts never count their slab ready and the last arriver's wait is short: the give-up path (NaN outputs, sticky status) on demand
    std::atomic<long long> enc_no_t16;          // DIINN_ENC_NO_T16 (TEST / A-B ONLY; default 0): 1 = small maps keep the split-K 3x3 kernel instead of the 1 x 16-pixel tile kernel (diinn_conv_t16.hip)
    std::atomic<long long> enc_wino4_split;     // DIINN_ENC_WINO4_SPLIT: F(4x4,3x3) kernel, split of the last round over the input channels: 0 never, 1 by the cost model (default), 2 whenever a round is partly filled
};

DiinnKnobs& diinn_knobs();                      // diinn_host.cpp
inline long long knob(const std::atomic<long long>& k) { return k.load(std::memory_order_relaxed); }
