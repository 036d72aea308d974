// fs2_pack.h - the PACKED row space of a lens-carrying contraction launch (fs2_gemm_w.hip, fs2_gemm_p.hip).
//
// Activations live in HBM as [B * S] rows, sequence b at rows b*S .. b*S + S - 1, of which the first lens[b] are valid
// (reference: the padded batches of dataset.py:collate_fn / utils/tools.py:pad_1D, masked after every sub-layer,
// transformer/Layers.py:25,28).  The bench batch is 83 % valid (LibriTTS buckets 46 %), but a 256-row tile of the padded space is
// skippable only when it lies entirely inside one sequence's tail - 2 % of the tiles at T = 925.  So the big contractions walk
// the packed space instead: row p = cu[b] + t (cu = exclusive prefix sum of lens; fs2_tile_map writes it behind the tile list)
// and every operand / result row is addressed through  padded(p) = b * S + (p - cu[b]).  A tile gathers 256 valid rows,
// possibly of two (or, for short sequences, several) neighbouring sequences; tap validity is judged against the row's OWN
// sequence (0 <= t + shift < lens[b]) exactly as the start of a sequence already was.  Padded rows of the result are written as
// zeros by the workgroups before they start (they are never computed).  Nothing else changes: tensors keep the padded layout.
#pragma once
#include <stdint.h>

// padded row of packed row pr (0 <= pr < P); cu: B + 1 prefix sums (in LDS); b: in/out hint, any value in [0, B)
__device__ __forceinline__ int fs2_packed_row(const int32_t* cu, int B, int S, int pr, int& b) {
    while (b > 0 && pr < cu[b]) --b;
    while (b + 1 < B && pr >= cu[b + 1]) ++b;
    return b * S + (pr - cu[b]);
}
