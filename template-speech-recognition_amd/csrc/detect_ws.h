// detect_ws.h -- interface between the host side of the MFMA bank scorer (detect_mfma.hip) and
// the wave-specialised kernel (detect_ws.hip).
#pragma once
#include "tsr_internal.h"

namespace tsr_ws {

constexpr int kWorkers = 8;    // MFMA waves per workgroup (one 32-column tile each)
constexpr int kThreads = (kWorkers + 1) * 64;  // + the helper wave
constexpr int kBuffers = 5;    // 32-frame tile buffers in flight
constexpr int kAhead = 3;      // the helper stages tile T+3 while the workers are around tile T
constexpr int kMaxSlots = 64;  // template fragments per workgroup (= tsr_fast::kMaxSlots)
constexpr int kMaxNKB = 12;    // K-blocks of 32 features a worker keeps in registers

struct SlotMeta {  // same 16 bytes as tsr_fast::SlotMeta
    float inv_scale;
    float cf;
    int L;
    int tf;
};

struct Args {
    const uint8_t *X;        // resident rows [P + slack][Dp], bytes in {0,1}
    int Dp;                  // resident row stride
    int SB;                  // LDS row stride (odd multiple of 16)
    int src_slots;           // 16-byte slots copied per frame
    int tile_bytes;          // bytes of one LDS tile buffer (multiple of 1024)
    int G, CT, nseg;
    const int32_t *seg;      // [3][nseg] segment plan (first output, end of outputs, end of rows)
    const void *Wl;          // limb fragments [G*CT][NKB][2 limbs][64 lanes] x 16 B (tsr_fast layout)
    const int32_t *col_slot; // [G*CT*32] slot of a column inside its group (-1 = padding)
    const int32_t *col_tau;  // [G*CT*32] tap of the column inside its template
    const SlotMeta *slots;   // [G][kMaxSlots]
    const int32_t *g_nslots; // [G]
    const int32_t *utt_end;  // [P + slack]
    int64_t P;
    int ring;                // ring length in frames (power of two)
    int lag;                 // tiles between a frame's tile and the tile that completes its sums
    int lmax1;               // Lmax - 1
    float *scores;           // MODE 0
    int32_t *ids;
    float *all;              // MODE 1: [M][P]
    float *raw_frag;         // MODE 2: [n_frag][P]
};

// nkb: K-blocks of 32 features (1..kMaxNKB).  Returns TSR_OK or a HIP error.
int launch(tsr_ctx *ctx, const Args &args, int nkb, bool flat, int grid, size_t lds);

}  // namespace tsr_ws
