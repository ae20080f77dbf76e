// placeholder: replaced by the parallel insert
#include "cafe_internal.h"
namespace cafe {
struct ParWorkspace {};
int par_insert(cafe_sketch *s, const int32_t *d_ids, const float *d_scores, int n, int32_t *d_ins, hipStream_t st) {
    s->counters[2]++;
    return launch_serial_insert(s, d_ids, d_scores, n, d_ins, 0, st);
}
void par_free(cafe_sketch *) {}
}
