// Storage of the inverse factor W = L^-1 of the polish's dual solve (lh_inverse.inc): rows in groups of eight; the rows of group g
// (8g .. 8g+7) are stored with 8(g+1) entries each -- the part right of the diagonal is kept ZERO -- at a stride of 8(g+1)+1 doubles.
#pragma once
__host__ __device__ inline int lhp_len(int g) { return 8 * (g + 1); }                   // entries of a row of group g
__host__ __device__ inline int lhp_grp(int g) { return 32 * g * (g + 1) + 8 * g; }        // offset of row 8g
__host__ __device__ inline int lhp_row(int b) { return lhp_grp(b >> 3) + (b & 7) * (lhp_len(b >> 3) + 1); }
__host__ __device__ inline int lhp_size(int rows) { return lhp_grp((rows + 7) >> 3); }    // doubles for `rows` rows
