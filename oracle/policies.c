/* oracle/policies.c -- TEST INFRASTRUCTURE ONLY (see cerebro_oracle.h).
 *
 * CPU restatement of the two top-k candidate policies of the reference that sit on a faiss::IndexFlatIP
 * (compiled out upstream, HAVE_FAISS undefined):
 *   Cerebro::faiss__naive_loopcandidate_generator   /root/reference/src/Cerebro.cpp:366-492
 *   Cerebro::faiss_clique_loopcandidate_generator   /root/reference/src/Cerebro.cpp:506-722
 * One call = one iteration of the while-loop body.  The index holds rows [0, l-150) (add step :415-433 / :558-582);
 * search(1, x, 5) is the top-5 inner products in descending order (:460 / :605).  faiss accumulates in fp32 in an
 * unspecified order; here a score is the fixed-tree fp64 sum of DESIGN.md 3 rounded once to float (the correctly
 * rounded fp32 inner product up to 1 double-rounding), ties ordered by larger index first.
 * Parity unpinned against faiss itself (dependency absent, policies never built upstream).                          */
#include <assert.h>
#include <stdlib.h>
#include <string.h>

#include "cerebro_oracle.h"

enum { START_ADDING_AFTER = 150, KNN = 5 };   /* Cerebro.cpp:375 / :515-516 */

static void search5(const float *db, int32_t D, int64_t ntotal, int64_t row, float *distances, int64_t *labels)
{
    double s[KNN];
    orc_scan_topk_f32(db, ntotal, D, db + (size_t)row * D, 1, KNN, s, labels);
    for (int g = 0; g < KNN; g++) distances[g] = (float)s[g];
}

int32_t orc_faiss_naive_tick(const float *db, int32_t D, int64_t l, orc_naive_state *st, orc_policy_loop *out)
{
    const int LOCALITY_THRESH = 12;          /* :376 */
    const float DOT_PROD_THRESH = 0.9f;      /* :377 */
    if (l - st->last_l < 3) return 0;        /* :403-407 "nothing new" */
    if (l > START_ADDING_AFTER) st->l_last_added = l - START_ADDING_AFTER;   /* :415-433; ntotal == l_last_added */
    const int64_t ntotal = st->l_last_added;
    float tmp_[8];
    int tmp_i[8];
    int n = 0, overflow = 0;
    for (int64_t l_i = st->last_l; l_i < l; l_i++) {       /* :441-473 */
        if (ntotal < 5) continue;                          /* :451 */
        float distances[KNN];
        int64_t labels[KNN];
        search5(db, D, ntotal, l_i, distances, labels);
        if (n < 8) { tmp_[n] = distances[0]; tmp_i[n] = (int)labels[0]; n++; } else overflow = 1;
    }
    int32_t found = 0;
    if (!overflow && n == 3 && tmp_[n - 1] > DOT_PROD_THRESH && abs(tmp_i[0] - tmp_i[1]) < LOCALITY_THRESH &&
        abs(tmp_i[0] - tmp_i[2]) < LOCALITY_THRESH) {      /* :476 */
        out->idx_curr = l - 1;                             /* :484 */
        out->idx_prev = tmp_i[2];
        out->score = (double)tmp_[2];
        found = 1;
    }
    st->last_l = l;                                        /* :488 */
    return found;
}

/* std::map<idx_t,int> retained (:539): keys kept ascending */
static int retained_find_duplicate(const orc_clique_state *st, int64_t label, int locality)
{
    for (int i = 0; i < st->n_retained; i++)
        if (st->key[i] - label < locality) return i;       /* :634 -- signed difference, no abs(): kept as written */
    return -1;
}

static void retained_insert(orc_clique_state *st, int64_t label)
{
    assert(st->n_retained < ORC_CLIQUE_MAX_RETAINED);
    int pos = 0;
    while (pos < st->n_retained && st->key[pos] < label) pos++;
    /* reaching here means no key k has k - label < LOCALITY, so label itself is not a key */
    memmove(&st->key[pos + 1], &st->key[pos], sizeof(int64_t) * (size_t)(st->n_retained - pos));
    memmove(&st->cnt[pos + 1], &st->cnt[pos], sizeof(int32_t) * (size_t)(st->n_retained - pos));
    st->key[pos] = label;
    st->cnt[pos] = 1;
    st->n_retained++;
}

int32_t orc_faiss_clique_tick(const float *db, int32_t D, int64_t l, orc_clique_state *st,
                              int (*rnd)(void *), void *rnd_arg, orc_policy_loop *out, int32_t max_out)
{
    const double DOT_PROD_THRESH = 0.85;     /* :517 (double: the float distance is promoted) */
    const int LOCALITY = 7;                  /* :518 */
    const int reset_every = 4;               /* :519 */
    if (l <= st->last_l) return 0;           /* :543-547 */
    if (l > START_ADDING_AFTER) st->l_last_added = l - START_ADDING_AFTER;   /* :558-582 */
    const int64_t ntotal = st->l_last_added;
    int32_t n_out = 0;
    for (int64_t l_i = st->last_l; l_i < l; l_i++) {       /* :589 */
        if (ntotal < KNN) break;                           /* :596 */
        float distances[KNN];
        int64_t labels[KNN];
        search5(db, D, ntotal, l_i, distances, labels);
        for (int g = 0; g < KNN; g++) {                    /* :625-650 */
            if ((double)distances[g] < DOT_PROD_THRESH) break;
            const int dup = retained_find_duplicate(st, labels[g], LOCALITY);
            if (dup >= 0) st->cnt[dup]++;
            else retained_insert(st, labels[g]);
        }
        if (st->n_retained > 0 && l_i % reset_every == 0) {   /* :653 */
            if (st->n_retained == 1) {                        /* :672-684: score is the constant 0.9 */
                if (n_out < max_out) { out[n_out].idx_curr = l - 1; out[n_out].idx_prev = st->key[0]; out[n_out].score = 0.9; }
                n_out++;
            } else {                                          /* :686-700: rand()-thinned */
                const int percent = (int)(100. / st->n_retained);
                for (int i = 0; i < st->n_retained; i++)
                    if (rnd(rnd_arg) % 100 < percent) {
                        if (n_out < max_out) { out[n_out].idx_curr = l - 1; out[n_out].idx_prev = st->key[i]; out[n_out].score = 0.9; }
                        n_out++;
                    }
            }
            st->n_retained = 0;                               /* :703 */
        }
    }
    st->last_l = l;                                           /* :710 */
    return n_out;
}
