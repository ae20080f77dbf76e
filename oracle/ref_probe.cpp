// ref_probe.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Builds the reference's own tricks/sketch.cpp (included from where it lies under /root/reference,
// never copied) into oracle/_ref/sklib_probe.so with its private state made readable, so that
// tests/golden/make_golden.py can dump every slot of the reference's tables after each call.
// Only meaningful in the build container; the GPU box uses the committed fixtures instead.
#include <bits/stdc++.h>
#define private public
#define main cafe_reference_main
#include REF_SKETCH_CPP
#undef main
#undef private

extern "C" {

void probe_dims(int *out) {  // n1, n2, lim, free-list length
    out[0] = ss->n1; out[1] = ss->n2; out[2] = ss->lim; out[3] = (int)ss->hot_id.size();
}

void probe_scalars(float *thr, int *ints, double *decay) {
    *thr = ss->Threshold;
    ints[0] = ss->num; ints[1] = ss->real_n; ints[2] = ss->t; ints[3] = (int)global_flip_bit;
    ints[4] = batch_num; ints[5] = 0;
    *decay = decay_importance;
}

// 1 when the LRU table was allocated below the main table: reset() breaks score ties by address
int probe_lru_first() { return (void *)ss->b2 < (void *)ss->b ? 1 : 0; }

void probe_dump(int *mval, float *mcnt, int *mdic, unsigned char *mflip, int *lval, float *lcnt,
                int *ldic, int *lt, unsigned char *lflip, unsigned *fifo) {
    for (int k = 0; k < ss->n1; ++k) {
        for (int i = 0; i < 4; ++i) {
            mval[k * 4 + i] = ss->b[k].val[i]; mcnt[k * 4 + i] = ss->b[k].cnt[i];
            mdic[k * 4 + i] = ss->b[k].dic[i];
        }
        mflip[k] = ss->b[k].flip_bit;
    }
    for (int k = 0; k < ss->n2; ++k) {
        for (int i = 0; i < 4; ++i) {
            lval[k * 4 + i] = ss->b2[k].val[i]; lcnt[k * 4 + i] = ss->b2[k].cnt[i];
            ldic[k * 4 + i] = ss->b2[k].dic[i]; lt[k * 4 + i] = ss->b2[k].t[i];
        }
        lflip[k] = ss->b2[k].flip_bit;
    }
    std::queue<uint32_t> q = ss->hot_id;
    for (size_t i = 0; !q.empty(); ++i) { fifo[i] = q.front(); q.pop(); }
}
}
