/* dlaf_c/desc.h -- matrix descriptor of the C interface.
 * Drop-in for the reference's include/dlaf_c/desc.h:16-26 (same field order and meaning, so a
 * caller compiled against either header passes the identical struct by value). */
#pragma once

struct DLAF_descriptor {
  int m;    /* rows of the global matrix */
  int n;    /* columns of the global matrix */
  int mb;   /* row block size (must equal nb for Cholesky) */
  int nb;   /* column block size */
  int isrc; /* process row owning the first block row */
  int jsrc; /* process column owning the first block column */
  int i;    /* first row of the sub-matrix: must be 0 (sub-matrices unsupported, as upstream) */
  int j;    /* first column of the sub-matrix: must be 0 */
  int ld;   /* leading dimension of the caller's local column-major array */
};
