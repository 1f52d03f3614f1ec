// grm_h5.cpp -- Kover HDF5 writer (dsk2kover's output side).  Placeholder until the
// libhdf5 dlopen() writer lands: fails loudly instead of writing nothing.
#include "../../include/grm_kmer.h"

extern "C" int grm_write_kover_h5(grm_matrix *m, const char *existing_h5_path, int gzip_level, int chunk_cols)
{
    (void)m; (void)existing_h5_path; (void)gzip_level; (void)chunk_cols;
    return GRM_ERR_UNSUPPORTED;
}
