// engine.hip -- the batch engine behind the C ABI (include/mdfri.h, "Batch engine"): planner + per-chunk launch sequence + GO
// heads for B proteins and several heads at once.  The counterpart of reference pipeline.py:476-481
// (Pool.map(build_align_contact_map)) followed by pipeline.py:292-319 (_run_prediction_loop); SURVEY.md section 8b's
// `mdf_cmap_batch` / `mdf_gcn_forward_batch`.  No kernels here: the stages are the device entry points of cmap.hip / gcn.hip,
// strung together in C++ so that a consumer of the header needs nothing else, and so that a short launch sequence can be
// replayed as one hipGraph.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <functional>
#include <numeric>
#include <thread>
#include <vector>

#include "common.h"

using namespace mdf;

// ---------------------------------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------------------------------
struct PlanChunk {
    int32_t p0, p1;
    int64_t rows, row_off_pos;
    int32_t segment;
    int64_t group_base;
    int32_t max_len;   // longest query of the chunk: sizes the contact-bit words of ITS rows
    // which aggregation kernel takes which protein of the chunk (mdfri.h mdf_agg_desc): by length alone -- the maps of the fused path are
    // binary --, once for the launches whose operand is cache-resident ([0]: layer 2) and once for the others ([1]: layer 3 and up)
    struct AggLists {
        int64_t plist_pos;          // position of the list of matrix-pipe proteins (chunk-local indices, class after class) in mdf_plan::agg_plist
        int64_t skip_pos;           // position (in int32 words of agg_plist) of the chunk's bitmap of 16-row groups owned by listed proteins
        int32_t n_mf[3];            // proteins per length class (mdf_agg_class); [0]: those whose layer 1 is made inside the layer-2 launch (mdf_agg_l1_fused)
        int32_t n_plain[3];         // [0] only: the listed proteins that run the plain kernel on H1 rows written by k_layer1; they follow the others in the list
        bool last_listed;           // the chunk's last protein is on the list (its workgroups zero the rows behind it)
        std::vector<int32_t> csr_seg;   // (first row, row count) pairs of the rows of the other proteins: the CSR gather
        std::vector<int32_t> l1_seg;    // [0] only: the rows k_layer1 makes (everything but the proteins counted in n_mf)
        int64_t l1_skip_pos;        // [0] only: bitmap of the 16-row groups of the proteins counted in n_mf
    } agg[2];
    int64_t tail_row0;          // first row behind the last protein's padded rows
};
struct PlanSegment {
    int32_t p0, p1;
    int64_t groups, grp_off_pos;
};
// proteins of consecutive chunks [c0, c1) that run through the LSTM together (heads with a language model)
struct LmGroup {
    int32_t c0, c1;
    std::vector<int64_t> bases;      // first row of each chunk inside the group's row space, + total
    std::vector<int32_t> lens_host;  // lengths in non-increasing order
    int32_t B, Lmax;
    size_t rows_pos, lens_pos;       // element offsets of this group's arrays in the plan's device mirror
};

struct mdf_plan {
    uint64_t serial = 0;
    int32_t B = 0, max_rows = 0, max_segment_groups = 0, max_len = 0;
    int64_t max_chunk_rows = 0, max_groups = 0;
    std::vector<int32_t> Lq, chunk_row_off, grp_off, agg_plist;   // Lq: lengths in PLAN order
    std::vector<int32_t> order;   // plan position -> index in the caller's batch; empty = the plan keeps the input order
    std::vector<PlanChunk> chunks;
    std::vector<PlanSegment> segments;
    // device mirror (created by the first engine call that uses the plan; one device per plan)
    mutable std::mutex mu;
    mutable int device = -1;
    mutable int32_t *d_chunk_row_off = nullptr, *d_grp_off = nullptr, *d_agg_plist = nullptr, *d_order = nullptr;
    // LSTM grouping, keyed by the engine parameters it depends on
    mutable int64_t lm_key[3] = {-1, -1, -1};
    mutable std::vector<LmGroup> lm_groups;
    mutable int64_t *d_lm_rows = nullptr;
    mutable int32_t *d_lm_lens = nullptr;
    // the mirrors are stream-ordered allocations, released in the order of the stream that used the plan last: freeing a plan while
    // the NEXT batch is running must not wait for the device (hipFree does)
    mutable hipStream_t last_stream = nullptr;
    mutable std::vector<int64_t> lm_rows_host;   // sources of the asynchronous uploads of the LSTM grouping: they live with the plan
    mutable std::vector<int32_t> lm_lens_host;
};

// stream-ordered allocation; a runtime without the pool falls back on the blocking allocator
static hipError_t plan_alloc(void **p, size_t bytes, hipStream_t st)
{
    if (hipMallocAsync(p, bytes, st) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
    return hipMalloc(p, bytes);
}

// release in stream order; a stream that is gone by now (or a runtime without the pool) falls back on the blocking free
static void plan_release(void *p, hipStream_t st)
{
    if (!p) return;
    if (hipFreeAsync(p, st) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
    }
}

static std::atomic<uint64_t> g_plan_serial{1};



// Which aggregation kernel takes which protein of a chunk, for one of the two descriptors (kind 0: in front of layer 2, kind 1: layer 3 and
// up): the list of matrix-pipe proteins (chunk-local indices, class after class; kind 0: those whose layer 1 is made inside the launch first,
// then the plain ones), the row segments / group bitmaps of what is left to the CSR gather and -- kind 0 -- to k_layer1.
struct AggListsHost {
    std::vector<int32_t> plist, csr_seg, l1_seg;
    std::vector<uint32_t> skip, l1_skip;
    int32_t n_mf[3] = {0, 0, 0}, n_plain[3] = {0, 0, 0};
    bool last_listed = false;
};
template <class ClassOf, class LenOf>
static void build_agg_lists(int kind, int32_t n, const int32_t *ro, int64_t R, ClassOf cls_of, LenOf len_of, AggListsHost &o)
{
    auto fused = [&](int32_t q) { return kind == 0 && cls_of(q) >= 0 && mdf_agg_l1_fused(len_of(q)) != 0; };
    o = AggListsHost();
    for (int pass = 0; pass < 2; ++pass)          // pass 0: layer 1 inside the launch (kind 0) / everything (kind 1); pass 1: the plain ones of kind 0
        for (int cls = 0; cls < 3; ++cls)
            for (int32_t q = 0; q < n; ++q)
                if (cls_of(q) == cls && (kind == 0 ? fused(q) == (pass == 0) : pass == 0)) o.plist.push_back(q), ++(pass == 0 ? o.n_mf : o.n_plain)[cls];
    const size_t words = (size_t)((R / GROUP_ROWS + 31) / 32);
    o.skip.assign(words, 0u);
    o.l1_skip.assign(kind == 0 ? words : 0, 0u);
    auto add_seg = [&](std::vector<int32_t> &seg, int32_t q) {
        if (!seg.empty() && seg[seg.size() - 2] + seg.back() == ro[q]) seg.back() += ro[q + 1] - ro[q];   // adjacent to the previous one: one launch
        else seg.push_back(ro[q]), seg.push_back(ro[q + 1] - ro[q]);
    };
    auto mark = [&](std::vector<uint32_t> &bits, int32_t q) {
        for (int32_t g = ro[q] / GROUP_ROWS; g < (ro[q] + len_of(q) + GROUP_ROWS - 1) / GROUP_ROWS; ++g) bits[(size_t)(g >> 5)] |= 1u << (g & 31);
    };
    for (int32_t q = 0; q < n; ++q) {
        if (cls_of(q) >= 0) mark(o.skip, q); else add_seg(o.csr_seg, q);
        if (kind == 0) { if (fused(q)) mark(o.l1_skip, q); else add_seg(o.l1_seg, q); }
    }
    o.last_listed = cls_of(n - 1) >= 0;
}

static void close_segment(mdf_plan *pl, const std::vector<int> &ids, int64_t groups)
{
    const PlanChunk &first = pl->chunks[ids.front()], &last = pl->chunks[ids.back()];
    PlanSegment sg{first.p0, last.p1, groups, (int64_t)pl->grp_off.size()};
    const size_t base = pl->grp_off.size();
    pl->grp_off.resize(base + (size_t)(last.p1 - first.p0) + 1);
    for (int ci : ids) {
        const PlanChunk &ch = pl->chunks[ci];
        const int32_t *ro = pl->chunk_row_off.data() + ch.row_off_pos;
        for (int32_t p = ch.p0; p < ch.p1; ++p) pl->grp_off[base + (size_t)(p - first.p0)] = (int32_t)(ch.group_base + ro[p - ch.p0] / GROUP_ROWS);
    }
    pl->grp_off[base + (size_t)(last.p1 - first.p0)] = (int32_t)groups;
    pl->segments.push_back(sg);
}

extern "C" int mdf_plan_create(const int32_t *Lq, int32_t B, int32_t max_rows, int32_t max_segment_groups, mdf_plan **out)
{
    return mdf_plan_create_ex(Lq, B, max_rows, max_segment_groups, 0u, out);
}

extern "C" const int32_t *mdf_plan_order(const mdf_plan *pl, int64_t *count)
{
    if (count) *count = pl ? (int64_t)pl->order.size() : 0;
    return pl && !pl->order.empty() ? pl->order.data() : nullptr;
}

extern "C" int32_t mdf_default_chunk_rows(void) { return MDF_DEFAULT_CHUNK_ROWS; }

extern "C" int mdf_plan_create_ex(const int32_t *Lq_in, int32_t B, int32_t max_rows, int32_t max_segment_groups, uint32_t flags, mdf_plan **out)
{
    MDF_REQUIRE(Lq_in && out, "plan_create: NULL argument");
    MDF_REQUIRE(B > 0, "plan_create: empty batch");
    if (max_rows <= 0) max_rows = MDF_DEFAULT_CHUNK_ROWS;
    if (max_segment_groups <= 0) max_segment_groups = 1 << 20;
    for (int32_t p = 0; p < B; ++p) MDF_REQUIRE(Lq_in[p] > 0, "plan_create: empty sequence in batch (protein %d)", p);
    auto *pl = new mdf_plan();
    pl->serial = g_plan_serial.fetch_add(1);
    pl->B = B;
    pl->max_rows = max_rows;
    pl->max_segment_groups = max_segment_groups;
    pl->Lq.assign(Lq_in, Lq_in + B);
    // visit the proteins shortest first (the reference sorts its work list by length, pipeline.py:529-533): proteins of like length share
    // chunks, and every chunk then holds few aggregation classes.  Stable, so equal lengths keep their input order; a batch that arrives
    // sorted keeps the identity (no order array, nothing permuted anywhere).
    if (!(flags & MDF_PLAN_KEEP_ORDER) && !std::is_sorted(Lq_in, Lq_in + B)) {
        pl->order.resize((size_t)B);
        for (int32_t p = 0; p < B; ++p) pl->order[(size_t)p] = p;
        std::stable_sort(pl->order.begin(), pl->order.end(), [&](int32_t a, int32_t b2) { return Lq_in[a] < Lq_in[b2]; });
        for (int32_t p = 0; p < B; ++p) pl->Lq[(size_t)p] = Lq_in[pl->order[(size_t)p]];
    }
    const int32_t *Lq = pl->Lq.data();
    int32_t p0 = 0;
    while (p0 < B) {
        int32_t p1 = p0, ml = 0;
        int64_t rows = 0;
        while (p1 < B) {
            const int64_t pad = ((int64_t)Lq[p1] + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
            if (p1 != p0 && rows + pad > max_rows) break;
            rows += pad;
            ml = std::max(ml, Lq[p1]);
            ++p1;
        }
        PlanChunk ch{};
        ch.p0 = p0;
        ch.p1 = p1;
        ch.row_off_pos = (int64_t)pl->chunk_row_off.size();
        ch.max_len = ml;
        pl->chunk_row_off.resize(pl->chunk_row_off.size() + (size_t)(p1 - p0) + 1);
        const int64_t R = mdf_layout_rows(Lq + p0, p1 - p0, pl->chunk_row_off.data() + ch.row_off_pos);
        if (R < 0) {
            delete pl;
            return (int)R;
        }
        ch.rows = R;
        {   // aggregation kernels by protein length
            const int32_t *ro = pl->chunk_row_off.data() + ch.row_off_pos;
            for (int kind = 0; kind < 2; ++kind) {
                PlanChunk::AggLists &al = ch.agg[kind];
                AggListsHost h;
                build_agg_lists(kind, p1 - p0, ro, R, [&](int32_t q) { return mdf_agg_class(Lq[p0 + q], kind == 0); }, [&](int32_t q) { return Lq[p0 + q]; }, h);
                al.plist_pos = (int64_t)pl->agg_plist.size();
                pl->agg_plist.insert(pl->agg_plist.end(), h.plist.begin(), h.plist.end());
                for (int c3 = 0; c3 < 3; ++c3) al.n_mf[c3] = h.n_mf[c3], al.n_plain[c3] = h.n_plain[c3];
                al.csr_seg = h.csr_seg;
                al.l1_seg = h.l1_seg;
                al.last_listed = h.last_listed;
                // bitmaps of the 16-row groups of the listed proteins (a gather / k_layer1 over all rows skips them: unsorted batches)
                al.skip_pos = (int64_t)pl->agg_plist.size();
                pl->agg_plist.insert(pl->agg_plist.end(), reinterpret_cast<const int32_t *>(h.skip.data()), reinterpret_cast<const int32_t *>(h.skip.data()) + h.skip.size());
                al.l1_skip_pos = (int64_t)pl->agg_plist.size();
                pl->agg_plist.insert(pl->agg_plist.end(), reinterpret_cast<const int32_t *>(h.l1_skip.data()), reinterpret_cast<const int32_t *>(h.l1_skip.data()) + h.l1_skip.size());
            }
            const int32_t last = p1 - 1 - p0;
            ch.tail_row0 = (int64_t)ro[last] + ((int64_t)Lq[p1 - 1] + GROUP_ROWS - 1) / GROUP_ROWS * GROUP_ROWS;
        }
        pl->chunks.push_back(ch);
        pl->max_chunk_rows = std::max(pl->max_chunk_rows, R);
        pl->max_len = std::max(pl->max_len, ml);
        p0 = p1;
    }
    // pooling segments: protein p's groups (GROUP_ROWS rows each) are [grp_off[p], grp_off[p+1]) inside its segment's partial array
    std::vector<int> cur;
    int64_t seg_groups = 0;
    for (int ci = 0; ci < (int)pl->chunks.size(); ++ci) {
        PlanChunk &ch = pl->chunks[ci];
        const int64_t g = ch.rows / GROUP_ROWS;
        if (!cur.empty() && seg_groups + g > max_segment_groups) {
            close_segment(pl, cur, seg_groups);
            pl->max_groups = std::max(pl->max_groups, seg_groups);
            cur.clear();
            seg_groups = 0;
        }
        ch.segment = (int32_t)pl->segments.size();
        ch.group_base = seg_groups;
        cur.push_back(ci);
        seg_groups += g;
    }
    close_segment(pl, cur, seg_groups);
    pl->max_groups = std::max(pl->max_groups, seg_groups);
    if (pl->max_groups >= 0x7fffffffLL) {
        delete pl;
        return fail(MDF_EINVAL, "plan_create: a pooling segment of %lld groups does not fit int32", (long long)seg_groups);
    }
    *out = pl;
    return MDF_OK;
}

extern "C" void mdf_plan_free(mdf_plan *pl)
{
    if (!pl) return;
    if (pl->d_chunk_row_off || pl->d_lm_rows) {
        DeviceGuard g(pl->device);
        plan_release(pl->d_chunk_row_off, pl->last_stream);   // one allocation holds both mirrors
        plan_release(pl->d_lm_rows, pl->last_stream);
    }
    delete pl;
}

extern "C" int32_t mdf_plan_num_proteins(const mdf_plan *pl) { return pl ? pl->B : 0; }
extern "C" int32_t mdf_plan_num_chunks(const mdf_plan *pl) { return pl ? (int32_t)pl->chunks.size() : 0; }
extern "C" int32_t mdf_plan_num_segments(const mdf_plan *pl) { return pl ? (int32_t)pl->segments.size() : 0; }
extern "C" int64_t mdf_plan_max_chunk_rows(const mdf_plan *pl) { return pl ? pl->max_chunk_rows : 0; }

extern "C" int mdf_plan_chunks(const mdf_plan *pl, int64_t *out)
{
    MDF_REQUIRE(pl && out, "plan_chunks: NULL argument");
    for (size_t c = 0; c < pl->chunks.size(); ++c) {
        const PlanChunk &ch = pl->chunks[c];
        int64_t *o = out + 6 * c;
        o[0] = ch.p0, o[1] = ch.p1, o[2] = ch.rows, o[3] = ch.row_off_pos, o[4] = ch.segment, o[5] = ch.group_base;
    }
    return MDF_OK;
}

extern "C" int mdf_plan_segments(const mdf_plan *pl, int64_t *out)
{
    MDF_REQUIRE(pl && out, "plan_segments: NULL argument");
    for (size_t s = 0; s < pl->segments.size(); ++s) {
        const PlanSegment &sg = pl->segments[s];
        int64_t *o = out + 4 * s;
        o[0] = sg.p0, o[1] = sg.p1, o[2] = sg.groups, o[3] = sg.grp_off_pos;
    }
    return MDF_OK;
}

extern "C" const int32_t *mdf_plan_chunk_row_off(const mdf_plan *pl, int64_t *count)
{
    if (!pl) return nullptr;
    if (count) *count = (int64_t)pl->chunk_row_off.size();
    return pl->chunk_row_off.data();
}

extern "C" const int32_t *mdf_plan_grp_off(const mdf_plan *pl, int64_t *count)
{
    if (!pl) return nullptr;
    if (count) *count = (int64_t)pl->grp_off.size();
    return pl->grp_off.data();
}

// the plan's two descriptor arrays on `device` (uploaded once, synchronously: a plan is made once per batch shape)
static int plan_mirror(const mdf_plan *pl, int device, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(pl->mu);
    const hipStream_t prev = pl->last_stream;
    pl->last_stream = st;
    if (pl->d_chunk_row_off && pl->device == device) return MDF_OK;
    if (pl->d_chunk_row_off || pl->d_lm_rows) {   // the plan moves to another device: drop the old mirror
        DeviceGuard g(pl->device);
        plan_release(pl->d_chunk_row_off, prev);
        plan_release(pl->d_lm_rows, prev);
        pl->d_chunk_row_off = nullptr;
        pl->d_lm_rows = nullptr;
        pl->lm_key[0] = -1;
    }
    const size_t n1 = pl->chunk_row_off.size(), n2 = pl->grp_off.size(), n3 = pl->agg_plist.size(), n4 = pl->order.size();
    int32_t *d = nullptr;
    MDF_HIP(plan_alloc(reinterpret_cast<void **>(&d), (n1 + n2 + n3 + n4) * 4 + 256, st));
    // the sources are the plan's own vectors (pageable: the runtime stages them before it returns); the copies sit in the stream in
    // front of the kernels that read the mirror
    if (hipMemcpyAsync(d, pl->chunk_row_off.data(), n1 * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d + n1, pl->grp_off.data(), n2 * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        (n3 && hipMemcpyAsync(d + n1 + n2, pl->agg_plist.data(), n3 * 4, hipMemcpyHostToDevice, st) != hipSuccess) ||
        (n4 && hipMemcpyAsync(d + n1 + n2 + n3, pl->order.data(), n4 * 4, hipMemcpyHostToDevice, st) != hipSuccess)) {
        plan_release(d, st);
        return fail(MDF_ENODEVICE, "plan: upload of the descriptor arrays failed");
    }
    pl->d_chunk_row_off = d;
    pl->d_grp_off = d + n1;
    pl->d_agg_plist = d + n1 + n2;
    pl->d_order = n4 ? d + n1 + n2 + n3 : nullptr;
    pl->device = device;
    return MDF_OK;
}

// Consecutive chunk ranges whose proteins run through the LSTM together: at most `cap` proteins and a time-major workspace
// (2 x (Lmax+1) x B x H floats) within `ws_bytes`.  The proteins are dealt evenly over the fewest such groups: an LSTM time
// step costs whole rounds of 256x256 tiles, so a small trailing group would cost as much as a full one.
static int plan_lm_groups(const mdf_plan *pl, int64_t lm_batch, int64_t ws_bytes, int64_t H, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(pl->mu);
    if (pl->lm_key[0] == lm_batch && pl->lm_key[1] == ws_bytes && pl->lm_key[2] == H && pl->d_lm_rows) return MDF_OK;
    const int64_t cap = std::min<int64_t>(lm_batch, 65535);
    const int nC = (int)pl->chunks.size();
    int64_t n_groups = std::max<int64_t>(1, (pl->B + cap - 1) / cap);
    std::vector<std::pair<int, int>> ranges;
    for (;;) {
        const int64_t target = (pl->B + n_groups - 1) / n_groups;
        ranges.clear();
        int c0 = 0;
        int64_t nb = 0, lmax = 0;
        for (int ci = 0; ci < nC; ++ci) {
            const PlanChunk &ch = pl->chunks[ci];
            const int64_t n = ch.p1 - ch.p0, l = ch.max_len;
            if (ci > c0 && (nb + n > cap || nb >= target || 8 * (std::max(lmax, l) + 1) * (nb + n) * H > ws_bytes)) {
                ranges.push_back({c0, ci});
                c0 = ci, nb = 0, lmax = 0;
            }
            nb += n;
            lmax = std::max(lmax, l);
        }
        ranges.push_back({c0, nC});
        if ((int64_t)ranges.size() <= n_groups || n_groups >= nC) break;
        n_groups = (int64_t)ranges.size();   // memory or chunk granularity forced more groups: re-balance for that count
    }
    pl->lm_groups.clear();
    std::vector<int64_t> all_rows;
    std::vector<int32_t> all_lens;
    for (auto &rg : ranges) {
        LmGroup g;
        g.c0 = rg.first, g.c1 = rg.second;
        g.bases.push_back(0);
        for (int ci = g.c0; ci < g.c1; ++ci) g.bases.push_back(g.bases.back() + pl->chunks[ci].rows);
        const int32_t P0 = pl->chunks[g.c0].p0, P1 = pl->chunks[g.c1 - 1].p1;
        std::vector<int64_t> prot_row((size_t)(P1 - P0));
        for (int ci = g.c0; ci < g.c1; ++ci) {
            const PlanChunk &ch = pl->chunks[ci];
            const int32_t *ro = pl->chunk_row_off.data() + ch.row_off_pos;
            for (int32_t p = ch.p0; p < ch.p1; ++p) prot_row[(size_t)(p - P0)] = g.bases[(size_t)(ci - g.c0)] + ro[p - ch.p0];
        }
        std::vector<int32_t> order((size_t)(P1 - P0));
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return pl->Lq[(size_t)(P0 + a)] > pl->Lq[(size_t)(P0 + b)]; });
        g.B = P1 - P0;
        g.rows_pos = all_rows.size();
        g.lens_pos = all_lens.size();
        for (int32_t k : order) {
            all_rows.push_back(prot_row[(size_t)k]);
            all_lens.push_back(pl->Lq[(size_t)(P0 + k)]);
            g.lens_host.push_back(pl->Lq[(size_t)(P0 + k)]);
        }
        g.Lmax = g.lens_host[0];
        pl->lm_groups.push_back(std::move(g));
    }
    plan_release(pl->d_lm_rows, st);
    pl->d_lm_rows = nullptr;
    char *d = nullptr;
    const size_t rb = align_up(all_rows.size() * 8, 256);
    MDF_HIP(plan_alloc(reinterpret_cast<void **>(&d), rb + all_lens.size() * 4 + 256, st));
    if (!pl->lm_rows_host.empty() && hipStreamSynchronize(st) != hipSuccess)   // a re-grouping: the previous upload may still be reading them
        return fail(MDF_ENODEVICE, "plan: stream synchronisation failed");
    pl->lm_rows_host.swap(all_rows);
    pl->lm_lens_host.swap(all_lens);
    if (hipMemcpyAsync(d, pl->lm_rows_host.data(), pl->lm_rows_host.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d + rb, pl->lm_lens_host.data(), pl->lm_lens_host.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess) {
        plan_release(d, st);
        return fail(MDF_ENODEVICE, "plan: upload of the LSTM group arrays failed");
    }
    pl->d_lm_rows = reinterpret_cast<int64_t *>(d);
    pl->d_lm_lens = reinterpret_cast<int32_t *>(d + rb);
    pl->lm_key[0] = lm_batch, pl->lm_key[1] = ws_bytes, pl->lm_key[2] = H;
    return MDF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    // grow-only; a reallocation frees the old block (hipFree waits for the device, so nothing in flight reads it)
    int grow(size_t need, uint64_t *generation, bool zero = false)
    {
        if (bytes >= need && p) return MDF_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        need = align_up(std::max<size_t>(need, 256), 256);
        MDF_HIP(hipMalloc(&p, need));
        if (zero) MDF_HIP(hipMemset(p, 0, need));
        bytes = need;
        if (generation) ++*generation;
        return MDF_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

// ---- plans that visit the proteins in another order than the batch stores them (mdf_plan_create: shortest first) ---------------------------
// The batch stays as the caller packed it; the engine derives descriptor arrays in PLAN order from it -- starts of the sequences, lengths,
// and (begin, end) pairs into the packed coordinates / alignments, which the contact stage reads with an offset stride of 2
// (mdf_cmap_csr_pairs_dev) -- and puts the pooled feature rows back into the caller's order in front of the GO heads, so scores, logits and
// every report are in input order.
__global__ void k_plan_order_desc(const int32_t *__restrict__ order, int B, const int32_t *__restrict__ seq_off, const int32_t *__restrict__ Lq,
                                  const int32_t *__restrict__ coord_off, const int32_t *__restrict__ aln_off, int32_t *__restrict__ seq_start,
                                  int32_t *__restrict__ Lq_o, int32_t *__restrict__ coord_pair, int32_t *__restrict__ aln_pair)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    const int src = order[p];
    seq_start[p] = seq_off[src];
    Lq_o[p] = Lq[src];
    if (coord_off) coord_pair[2 * p] = coord_off[src], coord_pair[2 * p + 1] = coord_off[src + 1];
    if (aln_off) aln_pair[2 * p] = aln_off[src], aln_pair[2 * p + 1] = aln_off[src + 1];
}
// dst[order[p]] = src[p], rows of `width` floats (float4 lanes)
__global__ __launch_bounds__(256) void k_rows_to_input_order(const float *__restrict__ src, float *__restrict__ dst, const int32_t *__restrict__ order, int width4)
{
    const float4 *s = reinterpret_cast<const float4 *>(src) + (size_t)blockIdx.x * width4;
    float4 *d = reinterpret_cast<float4 *>(dst) + (size_t)order[blockIdx.x] * width4;
    for (int c = threadIdx.x; c < width4; c += 256) d[c] = s[c];
}
static inline int plan_stride(const mdf_plan *pl) { return pl->order.empty() ? 1 : 2; }
// *view = the batch as the plan visits it (the caller's own descriptor when the plan keeps the input order)
static int batch_in_plan_order(DevBuf &perm, uint64_t *generation, const mdf_plan *pl, const mdf_batch_dev *b, hipStream_t st, mdf_batch_dev *view)
{
    *view = *b;
    if (pl->order.empty()) return MDF_OK;
    const size_t B = (size_t)pl->B;
    if (int rc = perm.grow(B * 6 * 4, generation)) return rc;
    int32_t *d = perm.as<int32_t>();
    hipLaunchKernelGGL(k_plan_order_desc, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, pl->d_order, (int)B, b->seq_off, b->Lq, b->coord_off, b->aln_off, d,
                       d + B, d + 2 * B, d + 4 * B);
    MDF_HIP(hipGetLastError());
    view->seq_off = d;
    view->Lq = d + B;
    if (b->coord_off) view->coord_off = d + 2 * B;
    if (b->aln_off) view->aln_off = d + 4 * B;
    return MDF_OK;
}
// pooled feature rows (plan order) -> the caller's order, into `scratch`; *out = where the GO head reads them
static int rows_to_input_order(const mdf_plan *pl, const DevBuf &rows, DevBuf &scratch, uint64_t *generation, size_t width, hipStream_t st, const float **out)
{
    *out = rows.as<float>();
    if (pl->order.empty()) return MDF_OK;
    MDF_REQUIRE(width % 4 == 0, "engine: feature rows of %zu floats", width);
    if (int rc = scratch.grow((size_t)pl->B * width * 4, generation)) return rc;
    hipLaunchKernelGGL(k_rows_to_input_order, dim3((unsigned)pl->B), dim3(256), 0, st, rows.as<float>(), scratch.as<float>(), pl->d_order, (int)(width / 4));
    MDF_HIP(hipGetLastError());
    *out = scratch.as<float>();
    return MDF_OK;
}

struct GraphEntry {
    std::vector<uint64_t> key;
    int seen = 0;            // calls with this key so far
    bool disabled = false;   // capture failed once: stay eager
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t last_use = 0;
};

// one slot of the host pipeline (mdf_engine_submit_alignments_host, near the end of this file)
struct mdf_host_slot {
    char *pin_in = nullptr, *pin_out = nullptr;
    size_t pin_in_bytes = 0, pin_out_bytes = 0;
    DevBuf d_in, d_out;
    mdf_plan *plan = nullptr;
    hipEvent_t ev_in = nullptr, ev_comp = nullptr, ev_done = nullptr;
    int64_t ticket = -1;        // -1: free
    int32_t B = 0;
    size_t in_bytes = 0, o_seq = 0, o_soff = 0, o_lq = 0, o_status = 0, o_bad = 0, out_flags = 0, out_bytes = 0;
    std::vector<size_t> s_off;  // per head: offset of its (B, T) block in d_out / pin_out
    mdf_batch_dev b{};
};

static void host_slots_release(mdf_engine *e);

struct mdf_engine {
    int device = 0;
    std::vector<mdf_model *> models;
    std::vector<mdf_lm *> lms;        // distinct attached language models
    std::vector<int> model_lm;        // per model: index into lms, or -1
    mdf_engine_config cfg{};
    bool want_lsum = false;           // some head has no language model: it takes the folded layer-1 operand
    int64_t lm_hidden_max = 0;
    // workspaces
    // the contact stage's outputs exist twice: while the GraphConv stacks of chunk c read one set, the contact stage of chunk
    // c+1 fills the other on a second stream (pipelined fused path); everything else uses set 0
    struct ContactSet {
        DevBuf rowptr, colidx, val, seq_idx, lsum, cws, dinv, blk, tiles;   // dinv / blk / tiles: operands of the matrix-pipe aggregation (mdf_agg_prepare_dev)
        hipEvent_t ready = nullptr, free = nullptr;
    } cs[2];
    hipStream_t aux = nullptr;        // low-priority stream of the pipelined contact stage
    hipEvent_t ev_fork = nullptr;
    bool pipeline_contact = false;
    DevBuf perm, pool_scratch;        // plans that order the batch: descriptors in plan order; the pooled rows on their way back to input order
    int last_set = 0;
    DevBuf gws, hws, seq_all, lm_ws, map_dev[2], flags;
    // the two slots of the host pipeline (mdf_engine_submit_alignments_host / mdf_engine_collect_host) and its three streams
    mdf_host_slot hslot[2];
    hipStream_t hs_in = nullptr, hs_comp = nullptr, hs_out = nullptr;
    int64_t next_ticket = 0;
    std::vector<DevBuf> partial, pooled, lm_h;
    int64_t rows_alloc = 0, nnz_cap = 0;
    int32_t len_alloc = 0;
    uint64_t generation = 0;          // bumped by every reallocation: captured graphs refer to old addresses
    std::mutex mu;
    // pinned staging + events of the dense-map path
    char *map_pin[2] = {nullptr, nullptr};
    size_t map_pin_bytes[2] = {0, 0};
    hipEvent_t map_ev[2] = {nullptr, nullptr}, map_up_ev[2] = {nullptr, nullptr};
    hipStream_t map_stream = nullptr;   // the maps cross PCIe on a stream of their own (highest priority: its own hardware queue), under the previous chunk's kernels
    // hipGraph cache (short launch sequences); captured on an engine-owned stream (the caller's may be the legacy default
    // stream, which cannot be captured), replayed on the caller's
    hipStream_t cap_stream = nullptr;
    std::vector<GraphEntry> graphs;
    uint64_t tick = 0;
    int64_t graph_launches = 0, eager_runs = 0;
    int64_t last_rows = 0;            // rows of the chunk whose CSR is in the buffers now
};

extern "C" int mdf_engine_create(mdf_model *const *models, int32_t n_models, int device, const mdf_engine_config *cfg, mdf_engine **out)
{
    MDF_REQUIRE(models && out && n_models > 0, "engine_create: at least one model is required");
    if (int rc = require_device()) return rc;
    auto *e = new mdf_engine();
    e->device = device;
    if (cfg) e->cfg = *cfg;
    mdf_engine_config &c = e->cfg;
    if (c.max_rows <= 0) c.max_rows = MDF_DEFAULT_CHUNK_ROWS;
    if (c.nnz_per_row <= 0) c.nnz_per_row = 40;
    if (!cfg) c.threshold = 6.0, c.generated_contacts = 2;
    if (c.max_segment_groups <= 0) c.max_segment_groups = 1 << 20;
    if (c.lm_batch <= 0) c.lm_batch = 16384;   // (round 6: one group of 10 000 proteins against two of 5 000: +1.1 % on the --lm line)
    if (c.lm_workspace_gib <= 0) c.lm_workspace_gib = 48.0;
    if (c.graph_max_chunks == 0) c.graph_max_chunks = 8;
    if (const char *g = getenv("MDFRI_ENGINE_GRAPH")) {   // developer knob: 0 = never replay graphs
        if (atoi(g) == 0) c.graph_max_chunks = -1;
    }
    for (int32_t k = 0; k < n_models; ++k) {
        mdf_model *m = models[k];
        if (!m || mdf_model_device(m) != device) {
            delete e;
            return fail(MDF_EINVAL, "engine_create: model %d is NULL or lives on another device", k);
        }
        e->models.push_back(m);
        int li = -1;
        if (mdf_model_lm_dim(m) > 0) {
            mdf_lm *lm = mdf_model_lm(m);
            if (!lm) {
                delete e;
                return fail(MDF_EINVAL, "engine_create: model %d has a language-model branch but no language model attached", k);
            }
            for (size_t i = 0; i < e->lms.size(); ++i)
                if (e->lms[i] == lm) li = (int)i;
            if (li < 0) {
                li = (int)e->lms.size();
                e->lms.push_back(lm);
                e->lm_hidden_max = std::max<int64_t>(e->lm_hidden_max, mdf_lm_hidden(lm));
            }
        } else {
            e->want_lsum = true;
        }
        e->model_lm.push_back(li);
    }
    // the pipelined contact stage (opt-in, cfg.pipeline_contact > 0; engines without a language model -- those encode whole LSTM
    // groups ahead of the contact stage): +0.8 % on the step, but the aggregation kernel loses 15 % next to the co-resident contact
    // kernels, so the default keeps everything on the caller's stream
    {
        const bool want = e->cfg.pipeline_contact > 0;
        if (want && e->lms.empty()) {
            DeviceGuard g(device);
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            bool ok = hipStreamCreateWithPriority(&e->aux, hipStreamNonBlocking, lo) == hipSuccess &&
                      hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess;
            for (auto &c : e->cs)
                ok = ok && hipEventCreateWithFlags(&c.ready, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&c.free, hipEventDisableTiming) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            e->pipeline_contact = ok;
        }
    }
    e->partial.resize(e->models.size());
    e->pooled.resize(e->models.size());
    e->lm_h.resize(e->lms.size());
    *out = e;
    return MDF_OK;
}

static void drop_graphs(mdf_engine *e)
{
    for (auto &g : e->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    e->graphs.clear();
}

extern "C" void mdf_engine_free(mdf_engine *e)
{
    if (!e) return;
    DeviceGuard g(e->device);
    (void)hipDeviceSynchronize();
    drop_graphs(e);
    host_slots_release(e);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    for (auto &c : e->cs) {
        for (DevBuf *b : {&c.rowptr, &c.colidx, &c.val, &c.seq_idx, &c.lsum, &c.cws, &c.dinv, &c.blk, &c.tiles}) b->release();
        if (c.ready) (void)hipEventDestroy(c.ready);
        if (c.free) (void)hipEventDestroy(c.free);
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->aux) (void)hipStreamDestroy(e->aux);
    e->perm.release();
    e->pool_scratch.release();
    if (e->map_stream) (void)hipStreamDestroy(e->map_stream);
    for (DevBuf *b : {&e->gws, &e->hws, &e->seq_all, &e->lm_ws, &e->map_dev[0], &e->map_dev[1], &e->flags}) b->release();
    for (auto &b : e->partial) b.release();
    for (auto &b : e->pooled) b.release();
    for (auto &b : e->lm_h) b.release();
    for (int i = 0; i < 2; ++i) {
        if (e->map_pin[i]) (void)hipHostFree(e->map_pin[i]);
        if (e->map_ev[i]) (void)hipEventDestroy(e->map_ev[i]);
        if (e->map_up_ev[i]) (void)hipEventDestroy(e->map_up_ev[i]);
    }
    delete e;
}

extern "C" int mdf_engine_set_nnz_per_row(mdf_engine *e, int32_t nnz_per_row)
{
    MDF_REQUIRE(e && nnz_per_row > 0, "engine_set_nnz_per_row: bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    e->cfg.nnz_per_row = nnz_per_row;
    e->rows_alloc = 0;   // the next call re-sizes the CSR arrays
    return MDF_OK;
}

extern "C" int64_t mdf_engine_nnz_capacity(const mdf_engine *e) { return e ? e->nnz_cap : 0; }
extern "C" int32_t mdf_engine_num_lms(const mdf_engine *e) { return e ? (int32_t)e->lms.size() : 0; }

// workspaces for chunks of up to `rows` rows, queries up to `max_len`, B proteins, segments of up to `groups` groups
static int ensure(mdf_engine *e, int64_t rows, int32_t B, int32_t max_len, int64_t groups)
{
    max_len = (max_len + 63) / 64 * 64;
    uint64_t *gen = &e->generation;
    if (rows > e->rows_alloc || max_len > e->len_alloc) {
        rows = std::max(rows, e->rows_alloc);
        max_len = std::max(max_len, e->len_alloc);
        const int64_t cap = std::max<int64_t>(rows * e->cfg.nnz_per_row, e->rows_alloc ? e->nnz_cap : 0);
        MDF_REQUIRE(cap < 0x7fffffffLL, "engine: %lld rows x %d entries per row exceed the int32 CSR; lower max_rows", (long long)rows, e->cfg.nnz_per_row);
        size_t gws = 0;
        for (mdf_model *m : e->models) gws = std::max(gws, mdf_gcn_workspace_bytes(m, rows));
        for (int k = 0; k < (e->pipeline_contact ? 2 : 1); ++k) {
            mdf_engine::ContactSet &c = e->cs[k];
            if (int rc = c.rowptr.grow((size_t)(rows + 1) * 4, gen)) return rc;
            if (int rc = c.colidx.grow((size_t)cap * 4, gen)) return rc;
            if (int rc = c.val.grow((size_t)cap * 4, gen)) return rc;
            if (int rc = c.seq_idx.grow((size_t)rows, gen)) return rc;
            if (int rc = c.lsum.grow((size_t)rows * 32 * 4, gen)) return rc;
            if (int rc = c.cws.grow(mdf_cmap_workspace_bytes(1 << 20, rows, max_len), gen)) return rc;
            if (int rc = c.dinv.grow((size_t)rows * 4, gen)) return rc;
            if (int rc = c.blk.grow((size_t)(rows / GROUP_ROWS + 1) * 32 * 8, gen)) return rc;   // (B, 32) x 64 bits; at most rows / 16 proteins in a chunk
            if (int rc = c.tiles.grow((size_t)rows * mdf_agg_tile_row_bytes(max_len), gen)) return rc;   // the contact bits as the kernel's byte tiles
        }
        if (int rc = e->gws.grow(gws, gen)) return rc;
        e->rows_alloc = rows;
        e->nnz_cap = cap;
        e->len_alloc = max_len;
    }
    size_t hws = 0;
    for (mdf_model *m : e->models) hws = std::max(hws, mdf_head_workspace_bytes(m, B));
    if (int rc = e->hws.grow(hws, gen)) return rc;
    for (size_t k = 0; k < e->models.size(); ++k) {
        const size_t feat = (size_t)mdf_model_feature_dim(e->models[k]);
        if (int rc = e->partial[k].grow((size_t)groups * feat * 4, gen)) return rc;
        if (int rc = e->pooled[k].grow((size_t)B * feat * 4, gen)) return rc;
    }
    return MDF_OK;
}

static int check_batch(const mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, bool need_coords)
{
    MDF_REQUIRE(e && pl && b, "engine: NULL argument");
    MDF_REQUIRE(b->B == pl->B, "engine: the batch holds %d proteins but the plan was made for %d", b->B, pl->B);
    MDF_REQUIRE(b->seqs && b->seq_off && b->Lq && b->status && b->bad, "engine: NULL pointer in the batch descriptor");
    if (need_coords)
        MDF_REQUIRE(b->coords && b->coord_off && b->q_aln && b->t_aln && b->aln_off, "engine: batch was packed without coordinates/alignments");
    return MDF_OK;
}

// per-chunk stages ---------------------------------------------------------------------------------------------------
static int encode_chunk(mdf_engine *, const mdf_plan *pl, const mdf_batch_dev *b, int ci, uint8_t *seq_ptr, hipStream_t st)
{
    const PlanChunk &ch = pl->chunks[(size_t)ci];
    return mdf_seq_encode_dev(b->seqs, b->seq_off + ch.p0, b->Lq + ch.p0, pl->d_chunk_row_off + ch.row_off_pos, ch.p1 - ch.p0, ch.rows, seq_ptr,
                              b->bad + ci, st);
}

// letter sums once per chunk (shared by every head without a language model), then the GraphConv stack of each head; the
// per-group partial sums land in the head's segment array
// Which aggregation kernel takes which protein of a chunk when that is not a function of the lengths alone (dense maps: a map that is
// not binary keeps the CSR gather): replaces the plan's lists for one chunk.
static const int32_t kNoSeg = 0;   // a non-NULL l1_seg of zero segments: "this descriptor names its layer-1 rows, and there are none"

struct AggOverride {
    struct Lists {
        const int32_t *d_plist = nullptr;
        const uint32_t *d_skip = nullptr, *d_l1_skip = nullptr;   // bitmaps of the 16-row groups of the listed proteins / of those counted in n_mf ([0])
        int32_t n_mf[3] = {0, 0, 0}, n_plain[3] = {0, 0, 0};
        bool last_listed = false;   // the chunk's last protein is on the list
        std::vector<int32_t> csr_seg, l1_seg;
    } k[2];   // [0]: layer 2, [1]: layer 3 and up
};

// The aggregation descriptors of one chunk (agg[0]: layer 2, agg[1]: layer 3 and up) + the launch that prepares what the matrix-pipe
// kernels read (dinv, populated-block bitmap).  *aggp = agg, or NULL when no protein of the chunk takes the matrix pipe.
static int chunk_agg_desc(mdf_engine *e, mdf_engine::ContactSet &c, const mdf_plan *pl, const mdf_batch_dev *b, const PlanChunk &ch, bool bits,
                          const AggOverride *ov, hipStream_t st, mdf_agg_desc agg[2], const mdf_agg_desc **aggp)
{
    memset(agg, 0, 2 * sizeof(mdf_agg_desc));
    *aggp = nullptr;
    int n_listed = 0;
    for (int kind = 0; kind < 2; ++kind)
        for (int c3 = 0; c3 < 3; ++c3) n_listed += ov ? ov->k[kind].n_mf[c3] + ov->k[kind].n_plain[c3] : ch.agg[kind].n_mf[c3] + ch.agg[kind].n_plain[c3];
    if (!(bits && n_listed > 0)) return MDF_OK;
    const uint64_t *masks = nullptr;
    const int32_t *counts = nullptr;
    int32_t W = 0;
    if (int rc = mdf_cmap_ws_view(c.cws.p, c.cws.bytes, ch.rows, ch.max_len, &masks, &W, &counts)) return rc;
    const int32_t *d_ro = pl->d_chunk_row_off + ch.row_off_pos, *d_lq = b->Lq + ch.p0;
    const int32_t Wt = mdf_agg_tile_row_bytes(ch.max_len);
    if (int rc = mdf_agg_prepare_dev(masks, W, counts, d_ro, d_lq, ch.p1 - ch.p0, ch.rows, c.dinv.as<float>(), c.blk.as<uint64_t>(), c.tiles.as<uint8_t>(), Wt, st)) return rc;
    for (int kind = 0; kind < 2; ++kind) {
        mdf_agg_desc &a = agg[kind];
        a.tiles = c.tiles.as<uint8_t>(), a.tile_row_bytes = Wt;
        a.masks = masks, a.W = W, a.dinv = c.dinv.as<float>(), a.blk = c.blk.as<uint64_t>(), a.row_off = d_ro, a.Lq = d_lq;
        a.tail_row0 = ch.tail_row0;
        bool last_listed;
        if (ov) {
            const AggOverride::Lists &l = ov->k[kind];
            a.plist = l.d_plist, a.csr_seg = l.csr_seg.data(), a.n_seg = (int32_t)(l.csr_seg.size() / 2), last_listed = l.last_listed;
            a.skip_groups = l.d_skip;
            for (int c3 = 0; c3 < 3; ++c3) a.n_mf[c3] = l.n_mf[c3], a.n_plain[c3] = l.n_plain[c3];
            if (kind == 0) a.l1_seg = l.l1_seg.empty() ? &kNoSeg : l.l1_seg.data(), a.n_l1_seg = (int32_t)(l.l1_seg.size() / 2), a.l1_skip = l.d_l1_skip;
        } else {
            const PlanChunk::AggLists &l = ch.agg[kind];
            a.plist = pl->d_agg_plist + l.plist_pos, a.csr_seg = l.csr_seg.data(), a.n_seg = (int32_t)(l.csr_seg.size() / 2), last_listed = l.last_listed;
            a.skip_groups = reinterpret_cast<const uint32_t *>(pl->d_agg_plist + l.skip_pos);
            for (int c3 = 0; c3 < 3; ++c3) a.n_mf[c3] = l.n_mf[c3], a.n_plain[c3] = l.n_plain[c3];
            if (kind == 0)
                a.l1_seg = l.l1_seg.empty() ? &kNoSeg : l.l1_seg.data(), a.n_l1_seg = (int32_t)(l.l1_seg.size() / 2), a.l1_skip = reinterpret_cast<const uint32_t *>(pl->d_agg_plist + l.l1_skip_pos);
        }
        // the rows behind the last protein: its workgroups zero them when it is on the matrix-pipe list; otherwise the gather segment of
        // that protein runs to the end of the rows and writes zeros there (empty CSR rows)
        a.tail_p = last_listed ? ch.p1 - ch.p0 - 1 : 0x7fffffff;
    }
    *aggp = agg;
    return MDF_OK;
}

// `bits`: the contact set's workspace holds this chunk's contact bits and degrees (the fused contact stage ran on it with (ch.rows,
// ch.max_len)): proteins of at most MDF_AGG_MAX_LEN residues then aggregate on the matrix pipe, the others through the CSR gather
static int gcn_chunk(mdf_engine *e, mdf_engine::ContactSet &c, const mdf_plan *pl, const mdf_batch_dev *b, const PlanChunk &ch, const uint8_t *seq_ptr,
                     const std::vector<const float *> &lm_h, bool have_lsum, bool bits, hipStream_t st, const AggOverride *ov = nullptr)
{
    if (!have_lsum && e->want_lsum)
        if (int rc = mdf_letter_sums_dev(seq_ptr, c.rowptr.as<int32_t>(), c.colidx.as<int32_t>(), c.val.as<float>(), ch.rows, c.lsum.as<float>(), st))
            return rc;
    mdf_agg_desc agg[2];   // [0]: layer 2 (operand cache-resident), [1]: layer 3 and up
    const mdf_agg_desc *aggp = nullptr;
    if (int rc = chunk_agg_desc(e, c, pl, b, ch, bits, ov, st, agg, &aggp)) return rc;
    for (size_t k = 0; k < e->models.size(); ++k) {
        mdf_model *m = e->models[k];
        float *part = e->partial[k].as<float>() + (size_t)ch.group_base * (size_t)mdf_model_feature_dim(m);
        int rc;
        if (e->model_lm[k] < 0)
            rc = mdf_gcn_embed_agg_dev(m, c.lsum.as<float>(), c.rowptr.as<int32_t>(), c.colidx.as<int32_t>(), c.val.as<float>(), ch.rows, aggp, part,
                                       e->gws.p, e->gws.bytes, st);
        else
            rc = mdf_gcn_embed_lm_agg_dev(m, seq_ptr, lm_h[(size_t)e->model_lm[k]], c.rowptr.as<int32_t>(), c.colidx.as<int32_t>(), c.val.as<float>(),
                                          ch.rows, aggp, part, e->gws.p, e->gws.bytes, st);
        if (rc) return rc;
    }
    return MDF_OK;
}

static int pool_segment(mdf_engine *e, const mdf_plan *pl, const PlanSegment &sg, hipStream_t st)
{
    for (size_t k = 0; k < e->models.size(); ++k) {
        mdf_model *m = e->models[k];
        if (int rc = mdf_gcn_pool_dev(m, e->partial[k].as<float>(), pl->d_grp_off + sg.grp_off_pos, sg.p1 - sg.p0,
                                      e->pooled[k].as<float>() + (size_t)sg.p0 * (size_t)mdf_model_feature_dim(m), st))
            return rc;
    }
    return MDF_OK;
}

// (have_lsum: the stage also produced the layer-1 letter sums; bits: it left the contact bits + degrees in the set's workspace)
using BuildCsr = std::function<int(int ci, const PlanChunk &ch, const uint8_t *seq_ptr, bool *have_lsum, bool *bits, const AggOverride **ov)>;

// Common driver: per chunk the residue indices are encoded, `build_csr` writes the adjacency (saying whether it also produced
// the layer-1 letter sums), then the GCN stack runs; segments are pooled as soon as their last chunk has been issued.  With a
// language model the chunks are taken a group of proteins at a time: all of them are encoded first, the LSTM runs over the
// whole group, then the per-chunk stages follow.
static int run_chunks(mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, const BuildCsr &build_csr, hipStream_t st)
{
    const int nC = (int)pl->chunks.size();
    std::vector<const float *> lm_ptr(e->lms.size(), nullptr);
    auto tail = [&](int ci, const uint8_t *seq_ptr) -> int {
        const PlanChunk &ch = pl->chunks[(size_t)ci];
        e->last_rows = ch.rows;
        e->last_set = 0;
        bool have_lsum = false, bits = false;
        const AggOverride *ov = nullptr;
        if (int rc = build_csr(ci, ch, seq_ptr, &have_lsum, &bits, &ov)) return rc;
        if (int rc = gcn_chunk(e, e->cs[0], pl, b, ch, seq_ptr, lm_ptr, have_lsum, bits, st, ov)) return rc;
        if (ci + 1 == nC || pl->chunks[(size_t)ci + 1].segment != ch.segment) return pool_segment(e, pl, pl->segments[(size_t)ch.segment], st);
        return MDF_OK;
    };
    if (e->lms.empty()) {
        for (int ci = 0; ci < nC; ++ci) {
            if (int rc = encode_chunk(e, pl, b, ci, e->cs[0].seq_idx.as<uint8_t>(), st)) return rc;
            if (int rc = tail(ci, e->cs[0].seq_idx.as<uint8_t>())) return rc;
        }
        return MDF_OK;
    }
    if (int rc = plan_lm_groups(pl, e->cfg.lm_batch, (int64_t)(e->cfg.lm_workspace_gib * 1073741824.0), e->lm_hidden_max, st)) return rc;
    // size the group-level buffers once, for the largest group (growing them between groups would stall the device)
    int64_t rows_max = 0;
    size_t ws_max = 0;
    for (const LmGroup &g : pl->lm_groups) {
        rows_max = std::max(rows_max, g.bases.back());
        for (mdf_lm *lm : e->lms) ws_max = std::max(ws_max, mdf_lm_workspace_bytes(lm, g.B, g.Lmax));
    }
    if (int rc = e->seq_all.grow((size_t)rows_max, &e->generation)) return rc;
    if (int rc = e->lm_ws.grow(ws_max, &e->generation)) return rc;
    for (size_t i = 0; i < e->lms.size(); ++i)
        if (int rc = e->lm_h[i].grow((size_t)rows_max * (size_t)mdf_lm_hidden(e->lms[i]) * 4, &e->generation, /*zero=*/true)) return rc;
    for (const LmGroup &g : pl->lm_groups) {
        uint8_t *seq_all = e->seq_all.as<uint8_t>();
        for (int ci = g.c0; ci < g.c1; ++ci)
            if (int rc = encode_chunk(e, pl, b, ci, seq_all + g.bases[(size_t)(ci - g.c0)], st)) return rc;
        for (size_t i = 0; i < e->lms.size(); ++i)
            if (int rc = mdf_lm_forward_dev(e->lms[i], seq_all, pl->d_lm_rows + g.rows_pos, pl->d_lm_lens + g.lens_pos, g.lens_host.data(), g.B,
                                            e->lm_h[i].as<float>(), e->lm_ws.p, e->lm_ws.bytes, st))
                return rc;
        for (int ci = g.c0; ci < g.c1; ++ci) {
            const int64_t base = g.bases[(size_t)(ci - g.c0)];
            for (size_t i = 0; i < e->lms.size(); ++i) lm_ptr[i] = e->lm_h[i].as<float>() + (size_t)base * (size_t)mdf_lm_hidden(e->lms[i]);
            if (int rc = tail(ci, seq_all + base)) return rc;
        }
    }
    return MDF_OK;
}

static int run_heads(mdf_engine *e, const mdf_plan *pl, float *const *scores, float *const *logits, hipStream_t st)
{
    // the scratch of an ordered plan is sized once, for the widest head (growing it between two heads would free what the first still reads)
    size_t wmax = 0;
    for (mdf_model *m : e->models) wmax = std::max(wmax, (size_t)mdf_model_feature_dim(m));
    if (!pl->order.empty())
        if (int rc = e->pool_scratch.grow((size_t)pl->B * wmax * 4, &e->generation)) return rc;
    for (size_t k = 0; k < e->models.size(); ++k) {
        MDF_REQUIRE(scores[k], "engine: scores[%zu] is NULL", k);
        const float *pooled = nullptr;   // rows in the CALLER's order: scores, logits and everything behind them need no further mapping
        if (int rc = rows_to_input_order(pl, e->pooled[k], e->pool_scratch, &e->generation, (size_t)mdf_model_feature_dim(e->models[k]), st, &pooled)) return rc;
        if (int rc = mdf_gcn_head_dev(e->models[k], pooled, pl->B, scores[k], logits ? logits[k] : nullptr, e->hws.p, e->hws.bytes, st)) return rc;
    }
    return MDF_OK;
}

// contact stage of chunk ci into contact set c: residue indices, then coordinates + alignment -> normalised CSR (+ the layer-1
// letter sums when some head takes the folded embedding)
static int contact_chunk(mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, int ci, mdf_engine::ContactSet &c, hipStream_t st)
{
    const PlanChunk &ch = pl->chunks[(size_t)ci];
    if (int rc = encode_chunk(e, pl, b, ci, c.seq_idx.as<uint8_t>(), st)) return rc;
    const int os = plan_stride(pl);
    return mdf_cmap_csr_pairs_dev(b->coords, b->coord_off + os * ch.p0, b->q_aln, b->t_aln, b->aln_off + os * ch.p0, os, b->Lq + ch.p0, pl->d_chunk_row_off + ch.row_off_pos,
                            ch.p1 - ch.p0, ch.rows, ch.max_len, e->cfg.threshold, e->cfg.generated_contacts, c.rowptr.as<int32_t>(), c.colidx.as<int32_t>(),
                            c.val.as<float>(), e->nnz_cap, b->status + 4 * ci, e->want_lsum ? c.seq_idx.as<uint8_t>() : nullptr,
                            e->want_lsum ? c.lsum.as<float>() : nullptr, c.cws.p, c.cws.bytes, st);
}

static int forward_alignments_eager(mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b_in, float *const *scores, float *const *logits,
                                    hipStream_t st, bool capturing = false)
{
    mdf_batch_dev view;   // the batch in the plan's order (the caller's own descriptor when the plan keeps the input order)
    if (int rc = batch_in_plan_order(e->perm, &e->generation, pl, b_in, st, &view)) return rc;
    const mdf_batch_dev *b = &view;
    if (!e->pipeline_contact) {
        BuildCsr build = [&](int ci, const PlanChunk &ch, const uint8_t *seq_ptr, bool *have_lsum, bool *bits, const AggOverride **) -> int {
            // contact stage: coordinates read once; the CSR fill also writes the layer-1 letter sums of the chunk
            *have_lsum = e->want_lsum;
            *bits = true;
            mdf_engine::ContactSet &c = e->cs[0];
            const int os = plan_stride(pl);
            return mdf_cmap_csr_pairs_dev(b->coords, b->coord_off + os * ch.p0, b->q_aln, b->t_aln, b->aln_off + os * ch.p0, os, b->Lq + ch.p0,
                                    pl->d_chunk_row_off + ch.row_off_pos, ch.p1 - ch.p0, ch.rows, ch.max_len, e->cfg.threshold, e->cfg.generated_contacts,
                                    c.rowptr.as<int32_t>(), c.colidx.as<int32_t>(), c.val.as<float>(), e->nnz_cap, b->status + 4 * ci,
                                    e->want_lsum ? seq_ptr : nullptr, e->want_lsum ? c.lsum.as<float>() : nullptr, c.cws.p, c.cws.bytes, st);
        };
        if (int rc = run_chunks(e, pl, b, build, st)) return rc;
        return run_heads(e, pl, scores, logits, st);
    }
    // Pipelined form (no language model): the contact stage of chunk c+1 runs on a second, low-priority stream while the GraphConv
    // stacks of chunk c hold the matrix pipe.  Its kernels are small and compute/latency-bound (27-60 VGPRs, next to nothing in
    // LDS, a few MB of traffic): they fit on the SIMDs NEXT to a resident GEMM workgroup (216 of 512 VGPRs per wave, two waves)
    // and, unlike the aggregation, do not compete for L2.  Two sets of contact outputs alternate; events order producer and
    // consumer; every piece of second-stream work is joined back into `st` before the call's last launch.
    const int nC = (int)pl->chunks.size();
    hipStream_t sc = e->aux;
    const std::vector<const float *> no_lm;
    MDF_HIP(hipEventRecord(e->ev_fork, st));          // whatever precedes this call on st (uploads, the previous batch) comes first
    MDF_HIP(hipStreamWaitEvent(sc, e->ev_fork, 0));
    if (int rc = contact_chunk(e, pl, b, 0, e->cs[0], sc)) return rc;
    MDF_HIP(hipEventRecord(e->cs[0].ready, sc));
    for (int ci = 0; ci < nC; ++ci) {
        mdf_engine::ContactSet &cur = e->cs[ci & 1];
        if (ci + 1 < nC) {
            mdf_engine::ContactSet &nxt = e->cs[(ci + 1) & 1];
            if (ci >= 1) MDF_HIP(hipStreamWaitEvent(sc, nxt.free, 0));   // chunk ci-1 has finished reading that set
            if (int rc = contact_chunk(e, pl, b, ci + 1, nxt, sc)) return rc;
            MDF_HIP(hipEventRecord(nxt.ready, sc));
        }
        const PlanChunk &ch = pl->chunks[(size_t)ci];
        MDF_HIP(hipStreamWaitEvent(st, cur.ready, 0));
        e->last_rows = ch.rows;
        e->last_set = ci & 1;
        if (int rc = gcn_chunk(e, cur, pl, b, ch, cur.seq_idx.as<uint8_t>(), no_lm, /*have_lsum=*/e->want_lsum, /*bits=*/true, st)) return rc;
        if (ci + 1 == nC || pl->chunks[(size_t)ci + 1].segment != ch.segment)
            if (int rc = pool_segment(e, pl, pl->segments[(size_t)ch.segment], st)) return rc;
        MDF_HIP(hipEventRecord(cur.free, st));
    }
    return run_heads(e, pl, scores, logits, st);
}

// ---- hipGraph replay of short launch sequences ------------------------------------------------------------------------
// A batch below one chunk is launch-bound: ~12 launches per head plus the contact stage, each a few microseconds of kernel
// behind a host-side dispatch.  The sequence is a pure function of (plan, input/flag/output pointers, engine buffers,
// engine parameters), so the third identical call captures it into a graph and later calls replay it with one launch.
static std::vector<uint64_t> graph_key(const mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, float *const *scores,
                                       float *const *logits, hipStream_t st)
{
    std::vector<uint64_t> k = {pl->serial, (uint64_t)(uintptr_t)b->seqs, (uint64_t)(uintptr_t)b->seq_off, (uint64_t)(uintptr_t)b->Lq,
                               (uint64_t)(uintptr_t)b->coords, (uint64_t)(uintptr_t)b->coord_off, (uint64_t)(uintptr_t)b->q_aln,
                               (uint64_t)(uintptr_t)b->t_aln, (uint64_t)(uintptr_t)b->aln_off, (uint64_t)(uintptr_t)b->status,
                               (uint64_t)(uintptr_t)b->bad, e->generation, (uint64_t)e->nnz_cap, (uint64_t)(uintptr_t)st,
                               // the plan's device mirror is re-created when the plan moves to another device: a captured graph embeds its address
                               (uint64_t)(uintptr_t)pl->d_chunk_row_off, (uint64_t)(uintptr_t)pl->d_lm_rows};
    uint64_t thr;
    memcpy(&thr, &e->cfg.threshold, 8);
    k.push_back(thr);
    k.push_back((uint64_t)e->cfg.generated_contacts);
    for (size_t i = 0; i < e->models.size(); ++i) {
        k.push_back((uint64_t)(uintptr_t)scores[i]);
        k.push_back((uint64_t)(uintptr_t)(logits ? logits[i] : nullptr));
    }
    return k;
}

static int forward_alignments_locked(mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, float *const *scores, float *const *logits,
                                     hipStream_t st)
{
    if (int rc = plan_mirror(pl, e->device, st)) return rc;
    if (int rc = ensure(e, pl->max_chunk_rows, pl->B, pl->max_len, pl->max_groups)) return rc;
    const bool graphable = e->cfg.graph_max_chunks > 0 && (int)pl->chunks.size() <= e->cfg.graph_max_chunks && e->lms.empty() && !timing_on();
    if (!graphable) {
        ++e->eager_runs;
        return forward_alignments_eager(e, pl, b, scores, logits, st);
    }
    const std::vector<uint64_t> key = graph_key(e, pl, b, scores, logits, st);
    GraphEntry *ge = nullptr;
    for (auto &g : e->graphs)
        if (g.key == key) ge = &g;
    if (!ge) {
        if (e->graphs.size() >= 16) {   // evict the least recently used entry
            size_t victim = 0;
            for (size_t i = 1; i < e->graphs.size(); ++i)
                if (e->graphs[i].last_use < e->graphs[victim].last_use) victim = i;
            if (e->graphs[victim].exec) (void)hipGraphExecDestroy(e->graphs[victim].exec);
            if (e->graphs[victim].graph) (void)hipGraphDestroy(e->graphs[victim].graph);
            e->graphs.erase(e->graphs.begin() + (long)victim);
        }
        e->graphs.push_back(GraphEntry());
        ge = &e->graphs.back();
        ge->key = key;
    }
    ge->last_use = ++e->tick;
    ++ge->seen;
    if (ge->exec) {
        if (hipGraphLaunch(ge->exec, st) == hipSuccess) {
            ++e->graph_launches;
            return MDF_OK;
        }
        (void)hipGetLastError();
        ge->disabled = true;   // fall through to the eager path, for good
        (void)hipGraphExecDestroy(ge->exec);
        ge->exec = nullptr;
    }
    if (ge->disabled || ge->seen < 3) {   // calls 1 and 2 run eagerly (lazy one-time initialisation happens outside any capture)
        ++e->eager_runs;
        return forward_alignments_eager(e, pl, b, scores, logits, st);
    }
    // third call: capture (nothing executes), instantiate, launch on the caller's stream
    if ((!e->cap_stream && hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) ||
        hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        ge->disabled = true;
        ++e->eager_runs;
        return forward_alignments_eager(e, pl, b, scores, logits, st);
    }
    const int rc = forward_alignments_eager(e, pl, b, scores, logits, e->cap_stream, /*capturing=*/true);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(e->cap_stream, &graph);
    if (rc != MDF_OK || ce != hipSuccess || !graph) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        ge->disabled = true;
        if (rc != MDF_OK) return rc;   // an argument error is an argument error, captured or not
        ++e->eager_runs;
        return forward_alignments_eager(e, pl, b, scores, logits, st);
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || hipGraphLaunch(exec, st) != hipSuccess) {
        (void)hipGetLastError();
        if (exec) (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
        ge->disabled = true;
        ++e->eager_runs;
        return forward_alignments_eager(e, pl, b, scores, logits, st);
    }
    ge->graph = graph;
    ge->exec = exec;
    ++e->graph_launches;
    return MDF_OK;
}

extern "C" int mdf_engine_forward_alignments(mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, float *const *scores,
                                             float *const *logits, void *stream)
{
    if (int rc = check_batch(e, pl, b, true)) return rc;
    MDF_REQUIRE(scores, "engine_forward_alignments: scores is NULL");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    MDF_HIP(g.err);
    return forward_alignments_locked(e, pl, b, scores, logits, static_cast<hipStream_t>(stream));
}

// Diagnostic: forward calls replayed as graphs / issued eagerly since the engine was made.
extern "C" int mdf_engine_graph_stats(const mdf_engine *e, int64_t *graph_launches, int64_t *eager_runs)
{
    MDF_REQUIRE(e, "engine_graph_stats: NULL engine");
    if (graph_launches) *graph_launches = e->graph_launches;
    if (eager_runs) *eager_runs = e->eager_runs;
    return MDF_OK;
}

extern "C" int64_t mdf_engine_last_chunk_nnz(mdf_engine *e, void *stream)
{
    MDF_REQUIRE(e, "engine_last_chunk_nnz: NULL engine");
    std::lock_guard<std::mutex> lk(e->mu);
    MDF_REQUIRE(e->last_rows > 0 && e->cs[e->last_set].rowptr.p, "engine_last_chunk_nnz: nothing has run yet");
    DeviceGuard g(e->device);
    MDF_HIP(g.err);
    int32_t nnz = 0;
    MDF_HIP(hipMemcpyAsync(&nnz, e->cs[e->last_set].rowptr.as<int32_t>() + e->last_rows, 4, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    MDF_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return nnz;
}

// ---- dense-map path ------------------------------------------------------------------------------------------------
// host threads that stage dense maps into pinned memory: up to 8
static int host_copy_threads() { return (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 8u); }

extern "C" int mdf_engine_forward_dense(mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, const void *const *cmaps_host, int cmap_dtype,
                                        float *const *scores, float *const *logits, void *stream)
{
    if (int rc = check_batch(e, pl, b, false)) return rc;
    MDF_REQUIRE(cmaps_host && scores, "engine_forward_dense: NULL argument");
    MDF_REQUIRE(cmap_dtype == MDF_DT_I32 || cmap_dtype == MDF_DT_F32, "engine_forward_dense: maps must be int32 or float32 (dtype code %d)", cmap_dtype);
    for (int32_t p = 0; p < pl->B; ++p) MDF_REQUIRE(cmaps_host[p], "engine_forward_dense: contact map of protein %d is NULL", p);
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    MDF_HIP(g.err);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = plan_mirror(pl, e->device, st)) return rc;
    if (int rc = ensure(e, pl->max_chunk_rows, pl->B, pl->max_len, pl->max_groups)) return rc;   // (max_len: the workspace also takes the contact bits)
    for (int i = 0; i < 2; ++i) {
        if (!e->map_ev[i]) MDF_HIP(hipEventCreateWithFlags(&e->map_ev[i], hipEventDisableTiming));
        if (!e->map_up_ev[i]) MDF_HIP(hipEventCreateWithFlags(&e->map_up_ev[i], hipEventDisableTiming));
    }
    if (!e->map_stream) {
        int lo = 0, hi = 0;
        MDF_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        MDF_HIP(hipStreamCreateWithPriority(&e->map_stream, hipStreamNonBlocking, hi));
    }
    if (int rc = e->flags.grow((size_t)pl->B * 4, &e->generation)) return rc;   // per-protein "binary" flags the dense stage writes on the device (the host lists decide here)
    mdf_batch_dev view;   // the batch in the plan's order; the host maps are looked up through the same order
    if (int rc = batch_in_plan_order(e->perm, &e->generation, pl, b, st, &view)) return rc;
    b = &view;
    auto map_of = [&](int32_t p) { return cmaps_host[pl->order.empty() ? (size_t)p : (size_t)pl->order[(size_t)p]]; };
    int parity = 0, pending = -1;   // pending: the slot of the chunk whose GraphConv launches have been issued but whose event is not recorded yet
    bool used[2] = {false, false};
    AggOverride aov;
    BuildCsr build = [&](int ci, const PlanChunk &ch, const uint8_t *, bool *have_lsum, bool *bits, const AggOverride **ov) -> int {
        *have_lsum = false;
        *bits = true;    // the contact bits are left in the workspace; a map that is not binary keeps the CSR gather (decided below, on the host)
        *ov = &aov;
        const int32_t Bc = ch.p1 - ch.p0;
        // pack the chunk's maps + their element offsets into pinned memory: [offsets (Bc x int64) | maps]
        size_t elems = 0;
        for (int32_t p = ch.p0; p < ch.p1; ++p) elems += (size_t)pl->Lq[(size_t)p] * (size_t)pl->Lq[(size_t)p];
        // pinned block: [offsets (Bc x int64) | maps | the two lists of the proteins the matrix-pipe aggregation takes (<= 2 Bc x int32)]
        const size_t o_maps = align_up((size_t)Bc * 8, 256), o_plist = align_up(o_maps + elems * 4, 256), skip_words = (size_t)((ch.rows / GROUP_ROWS + 31) / 32),
                     o_skip = align_up(o_plist + (size_t)Bc * 8, 256), total = o_skip + 3 * skip_words * 4;   // (two lists + three group bitmaps, see below)
        const int s = parity;
        parity ^= 1;
        // A slot holds the maps AND the aggregation lists (plist / skip bitmaps) that every GraphConv launch of its chunk reads, for
        // every head and layer: it may be overwritten only once those launches are done.  run_chunks issues gcn_chunk(ci - 1) between
        // build(ci - 1) and build(ci), so the event of the previous chunk's slot is recorded HERE, behind its last launch (ADVICE r4:
        // recorded straight after the CSR conversion it guarded the maps only).
        if (pending >= 0) {
            MDF_HIP(hipEventRecord(e->map_ev[pending], st));
            used[pending] = true;
            pending = -1;
        }
        if (used[s]) MDF_HIP(hipEventSynchronize(e->map_ev[s]));   // every kernel that read this slot two chunks ago is done
        if (e->map_pin_bytes[s] < total) {
            if (e->map_pin[s]) (void)hipHostFree(e->map_pin[s]);
            e->map_pin[s] = nullptr;
            e->map_pin_bytes[s] = 0;
            MDF_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->map_pin[s]), total + total / 4, hipHostMallocDefault));
            e->map_pin_bytes[s] = total + total / 4;
        }
        if (int rc = e->map_dev[s].grow(total + total / 4, &e->generation)) return rc;
        int64_t *offs = reinterpret_cast<int64_t *>(e->map_pin[s]);
        char *dst = e->map_pin[s] + o_maps;
        int64_t nnz_needed = ch.rows, off = 0;
        for (int32_t p = ch.p0; p < ch.p1; ++p) {
            offs[p - ch.p0] = off;
            off += (int64_t)pl->Lq[(size_t)p] * (int64_t)pl->Lq[(size_t)p];
        }
        // copy + count the non-zeros of every map: 128 MiB per chunk at L = 512, two passes of one host core would take ten times
        // as long as the device needs for the chunk -- the proteins are dealt to a few host threads
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)host_copy_threads(), (int64_t)Bc, (int64_t)(elems >> 20) + 1}));
        std::vector<int64_t> nz_part((size_t)nt, 0);
        std::vector<uint8_t> other((size_t)Bc, 0);   // the map holds an entry off the diagonal that is neither 0 nor 1: not a binary contact map
        auto work = [&](int k) {
            int64_t nz = 0;
            for (int32_t p = ch.p0 + k; p < ch.p1; p += nt) {
                const size_t Lp = (size_t)pl->Lq[(size_t)p], n = Lp * Lp;
                char *to = dst + (size_t)offs[p - ch.p0] * 4;
                memcpy(to, map_of(p), n * 4);
                const uint32_t *w = reinterpret_cast<const uint32_t *>(to);
                uint32_t odd = 0;
                if (cmap_dtype == MDF_DT_I32) {
                    for (size_t i = 0; i < n; ++i) nz += w[i] != 0, odd |= w[i] > 1u;
                    if (odd)   // (rare) tell an odd diagonal -- the kernels force the diagonal to 1 whatever it holds -- from an odd entry elsewhere
                        for (size_t i = 0, hits = 0; i < n && !hits; ++i)
                            if (w[i] > 1u && i % (Lp + 1) != 0) other[(size_t)(p - ch.p0)] = 1, hits = 1;
                } else {
                    for (size_t i = 0; i < n; ++i) nz += (w[i] << 1) != 0, odd |= (uint32_t)((w[i] << 1) != 0 && w[i] != 0x3f800000u);   // +0.0 and -0.0 are zeros
                    if (odd)
                        for (size_t i = 0, hits = 0; i < n && !hits; ++i)
                            if ((w[i] << 1) != 0 && w[i] != 0x3f800000u && i % (Lp + 1) != 0) other[(size_t)(p - ch.p0)] = 1, hits = 1;
                }
            }
            nz_part[(size_t)k] = nz;
        };
        {
            std::vector<std::thread> pool;
            for (int k = 1; k < nt; ++k) pool.emplace_back(work, k);
            work(0);
            for (auto &th : pool) th.join();
        }
        for (int k = 0; k < nt; ++k) nnz_needed += nz_part[(size_t)k];
        if (nnz_needed > e->nnz_cap) {   // a denser chunk than the CSR arrays hold: grow them (hipFree waits for the device)
            MDF_REQUIRE(nnz_needed < 0x7fffffffLL, "engine_forward_dense: a chunk needs %lld CSR entries; lower max_rows", (long long)nnz_needed);
            // EVERY live contact set grows with the shared capacity: a later pipelined forward_alignments hands e->nnz_cap to both
            for (int k = 0; k < (e->pipeline_contact ? 2 : 1); ++k) {
                if (int rc = e->cs[k].colidx.grow((size_t)nnz_needed * 4, &e->generation)) return rc;
                if (int rc = e->cs[k].val.grow((size_t)nnz_needed * 4, &e->generation)) return rc;
            }
            e->nnz_cap = nnz_needed;
        }
        // aggregation kernel per protein: binary map and at most MDF_AGG_MAX_LEN residues -> the matrix pipe; the rest -> CSR gather segments
        {
            int32_t *plist = reinterpret_cast<int32_t *>(e->map_pin[s] + o_plist);
            const int32_t *ro = pl->chunk_row_off.data() + ch.row_off_pos;
            int32_t n_listed = 0;
            for (int kind = 0; kind < 2; ++kind) {
                AggOverride::Lists &l = aov.k[kind];
                AggListsHost h;
                build_agg_lists(kind, Bc, ro, ch.rows, [&](int32_t q) { return other[(size_t)q] ? -1 : mdf_agg_class(pl->Lq[(size_t)(ch.p0 + q)], kind == 0); },
                                [&](int32_t q) { return pl->Lq[(size_t)(ch.p0 + q)]; }, h);
                l.d_plist = reinterpret_cast<const int32_t *>(e->map_dev[s].as<char>() + o_plist) + n_listed;
                memcpy(plist + n_listed, h.plist.data(), h.plist.size() * 4);
                n_listed += (int32_t)h.plist.size();
                for (int c3 = 0; c3 < 3; ++c3) l.n_mf[c3] = h.n_mf[c3], l.n_plain[c3] = h.n_plain[c3];
                l.csr_seg = h.csr_seg;
                l.l1_seg = h.l1_seg;
                l.last_listed = h.last_listed;
                // three group bitmaps in the slot: [0] layer-2 listed, [1] layer-3 listed, [2] layer-2 proteins whose layer 1 is made inside the launch
                uint32_t *bits = reinterpret_cast<uint32_t *>(e->map_pin[s] + o_skip);
                memcpy(bits + (size_t)kind * skip_words, h.skip.data(), skip_words * 4);
                l.d_skip = reinterpret_cast<const uint32_t *>(e->map_dev[s].as<char>() + o_skip) + (size_t)kind * skip_words;
                if (kind == 0) {
                    memcpy(bits + 2 * skip_words, h.l1_skip.data(), skip_words * 4);
                    l.d_l1_skip = reinterpret_cast<const uint32_t *>(e->map_dev[s].as<char>() + o_skip) + 2 * skip_words;
                }
            }
        }
        char *d = e->map_dev[s].as<char>();
        // the device slot is free (map_ev[s] was waited for above): the upload overlaps whatever the compute stream is still doing
        MDF_HIP(hipMemcpyAsync(d, e->map_pin[s], total, hipMemcpyHostToDevice, e->map_stream));
        MDF_HIP(hipEventRecord(e->map_up_ev[s], e->map_stream));
        MDF_HIP(hipStreamWaitEvent(st, e->map_up_ev[s], 0));
        const int rc = mdf_dense_to_csr_masks_dev(d + o_maps, cmap_dtype, reinterpret_cast<const int64_t *>(d), b->Lq + ch.p0,
                                                  pl->d_chunk_row_off + ch.row_off_pos, Bc, ch.rows, ch.max_len, e->cs[0].rowptr.as<int32_t>(),
                                                  e->cs[0].colidx.as<int32_t>(), e->cs[0].val.as<float>(), e->nnz_cap, b->status + 4 * ci,
                                                  e->flags.as<int32_t>(), e->cs[0].cws.p, e->cs[0].cws.bytes, st);
        if (rc) return rc;
        pending = s;
        return MDF_OK;
    };
    ++e->eager_runs;
    if (int rc = run_chunks(e, pl, b, build, st)) return rc;
    if (int rc = run_heads(e, pl, scores, logits, st)) return rc;
    MDF_HIP(hipStreamSynchronize(st));
    return MDF_OK;
}

// ---- validation ------------------------------------------------------------------------------------------------------
static int check_locked(const mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, hipStream_t st, int64_t *info)
{
    const size_t nC = pl->chunks.size();
    std::vector<int64_t> bad(nC);
    std::vector<int32_t> status(nC * 4);
    MDF_HIP(hipMemcpyAsync(bad.data(), b->bad, nC * 8, hipMemcpyDeviceToHost, st));
    MDF_HIP(hipMemcpyAsync(status.data(), b->status, nC * 16, hipMemcpyDeviceToHost, st));
    MDF_HIP(hipStreamSynchronize(st));
    if (info) info[0] = info[1] = info[2] = info[3] = -1;
    // chunks hold consecutive proteins, so the first flagged chunk carries the first invalid byte of the whole batch: what the
    // reference's serial loop would have hit first (predict.pyx:36-46)
    for (size_t ci = 0; ci < nC; ++ci) {
        if (bad[ci] != -1) {
            int64_t p = pl->chunks[ci].p0 + (bad[ci] >> 32), pos = bad[ci] & 0xffffffffLL;
            if (!pl->order.empty()) {
                // the plan visits the proteins shortest first, so the flags name the first invalid residue in THAT order; the report is the first one
                // in the caller's order: look for it in the sequences themselves (error path only: one copy of the packed sequences back)
                const size_t B = (size_t)pl->B;
                std::vector<int32_t> soff(B + 1), lq(B);
                MDF_HIP(hipMemcpyAsync(soff.data(), b->seq_off, (B + 1) * 4, hipMemcpyDeviceToHost, st));
                MDF_HIP(hipMemcpyAsync(lq.data(), b->Lq, B * 4, hipMemcpyDeviceToHost, st));
                MDF_HIP(hipStreamSynchronize(st));
                std::vector<char> sq((size_t)std::max(soff[B], 1));
                MDF_HIP(hipMemcpyAsync(sq.data(), b->seqs, (size_t)soff[B], hipMemcpyDeviceToHost, st));
                MDF_HIP(hipStreamSynchronize(st));
                bool ok[256] = {false};
                for (const char *a = "-DGULNTKHYWCPVSOIEFXQABZRM"; *a; ++a) ok[(unsigned char)*a] = true;   // the residue alphabet of predict.pyx:26
                p = pl->order[(size_t)p];
                for (size_t q = 0, found = 0; q < B && !found; ++q)
                    for (int32_t i = 0; i < lq[q]; ++i)
                        if (!ok[(unsigned char)sq[(size_t)soff[q] + (size_t)i]]) {
                            p = (int64_t)q, pos = i, found = 1;
                            break;
                        }
            }
            if (info) info[0] = p, info[1] = pos;
            return fail(MDF_EBADCHAR, "Invalid character in sequence: protein %lld, position %lld", (long long)p, (long long)pos);
        }
    }
    int32_t too_long = 0, need = 0;
    bool overflow = false;
    for (size_t ci = 0; ci < nC; ++ci) {
        too_long = std::max(too_long, status[ci * 4 + 2]);
        if (status[ci * 4]) overflow = true, need = std::max(need, status[ci * 4 + 1]);
    }
    if (too_long) {
        if (info) info[2] = too_long;
        return fail(MDF_EINVAL, "a query of length %d exceeds the max_len the contact stage was given", too_long);
    }
    if (overflow) {
        if (info) info[3] = need;
        return fail(MDF_ECAPACITY, "CSR capacity %lld too small (a chunk needs %d); raise nnz_per_row", (long long)(e ? e->nnz_cap : 0), need);
    }
    return MDF_OK;
}

extern "C" int mdf_engine_check(mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, void *stream, int64_t info[4])
{
    if (int rc = check_batch(e, pl, b, false)) return rc;
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    MDF_HIP(g.err);
    return check_locked(e, pl, b, static_cast<hipStream_t>(stream), info);
}

// ---- sequence-only CNN models: the batched counterpart of the reference's CNN loop over the unaligned queries --------------
// (pipeline.py:600-648, `_run_prediction_loop(predictor=cnn, ...)`): residue indices chunk by chunk, conv + max pool per head
// into one (B, C) array, then the output layer ONCE per head over all proteins (a 2 048-protein chunk is too few rows to fill
// the GEMM's 256-row tiles on 256 CUs).
struct mdf_seq_engine {
    int device = 0;
    std::vector<mdf_cnn *> models;
    DevBuf seq_idx, ws, perm, pool_scratch;
    std::vector<DevBuf> pooled;
    std::mutex mu;
};

extern "C" int mdf_seq_engine_create(mdf_cnn *const *models, int32_t n_models, int device, mdf_seq_engine **out)
{
    MDF_REQUIRE(models && out && n_models > 0, "seq_engine_create: at least one model is required");
    if (int rc = require_device()) return rc;
    auto *e = new mdf_seq_engine();
    e->device = device;
    for (int32_t k = 0; k < n_models; ++k) {
        if (!models[k]) {
            delete e;
            return fail(MDF_EINVAL, "seq_engine_create: model %d is NULL", k);
        }
        e->models.push_back(models[k]);
    }
    e->pooled.resize(e->models.size());
    *out = e;
    return MDF_OK;
}

extern "C" void mdf_seq_engine_free(mdf_seq_engine *e)
{
    if (!e) return;
    DeviceGuard g(e->device);
    (void)hipDeviceSynchronize();
    e->seq_idx.release();
    e->ws.release();
    e->perm.release();
    e->pool_scratch.release();
    for (auto &b : e->pooled) b.release();
    delete e;
}

extern "C" int mdf_seq_engine_forward(mdf_seq_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, float *const *scores, void *stream)
{
    MDF_REQUIRE(e && pl && b && scores, "seq_engine_forward: NULL argument");
    MDF_REQUIRE(b->B == pl->B, "seq_engine_forward: the batch holds %d proteins but the plan was made for %d", b->B, pl->B);
    MDF_REQUIRE(b->seqs && b->seq_off && b->Lq && b->bad, "seq_engine_forward: NULL pointer in the batch descriptor");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    MDF_HIP(g.err);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = plan_mirror(pl, e->device, st)) return rc;
    mdf_batch_dev view;   // the batch in the plan's order
    if (int rc = batch_in_plan_order(e->perm, nullptr, pl, b, st, &view)) return rc;
    b = &view;
    const int64_t rows = pl->max_chunk_rows;
    if (int rc = e->seq_idx.grow((size_t)rows, nullptr)) return rc;
    if (int rc = e->ws.grow((size_t)(rows / GROUP_ROWS + 2) * 4 + 512, nullptr)) return rc;
    std::vector<int> cpad(e->models.size());
    for (size_t k = 0; k < e->models.size(); ++k) {
        cpad[k] = mdf_cnn_padded_channels(e->models[k]);
        if (int rc = e->pooled[k].grow((size_t)pl->B * (size_t)cpad[k] * 4, nullptr)) return rc;
    }
    for (size_t ci = 0; ci < pl->chunks.size(); ++ci) {
        const PlanChunk &ch = pl->chunks[ci];
        const int32_t *ro = pl->d_chunk_row_off + ch.row_off_pos;
        if (int rc = mdf_seq_encode_dev(b->seqs, b->seq_off + ch.p0, b->Lq + ch.p0, ro, ch.p1 - ch.p0, ch.rows, e->seq_idx.as<uint8_t>(), b->bad + ci, st))
            return rc;
        for (size_t k = 0; k < e->models.size(); ++k)
            if (int rc = mdf_cnn_pool_dev(e->models[k], e->seq_idx.as<uint8_t>(), b->Lq + ch.p0, ro, ch.p1 - ch.p0, ch.rows,
                                          e->pooled[k].as<float>() + (size_t)ch.p0 * (size_t)cpad[k], e->ws.p, e->ws.bytes, st))
                return rc;
    }
    size_t wmax = 0;
    for (int c : cpad) wmax = std::max(wmax, (size_t)c);
    if (!pl->order.empty())
        if (int rc = e->pool_scratch.grow((size_t)pl->B * wmax * 4, nullptr)) return rc;
    for (size_t k = 0; k < e->models.size(); ++k) {
        MDF_REQUIRE(scores[k], "seq_engine_forward: scores[%zu] is NULL", k);
        const float *pooled = nullptr;   // the pooled rows in the caller's order
        if (int rc = rows_to_input_order(pl, e->pooled[k], e->pool_scratch, nullptr, (size_t)cpad[k], st, &pooled)) return rc;
        if (int rc = mdf_cnn_head_dev(e->models[k], pooled, pl->B, scores[k], st)) return rc;
    }
    return MDF_OK;
}

extern "C" int mdf_seq_engine_check(mdf_seq_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, void *stream, int64_t info[4])
{
    MDF_REQUIRE(e && pl && b && b->bad && b->status, "seq_engine_check: NULL argument");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    MDF_HIP(g.err);
    return check_locked(nullptr, pl, b, static_cast<hipStream_t>(stream), info);
}

// ---- language-model features (inspection) ------------------------------------------------------------------------------
extern "C" int mdf_engine_lm_features_host(mdf_engine *e, const mdf_plan *pl, const mdf_batch_dev *b, int32_t which, float *out, void *stream)
{
    if (int rc = check_batch(e, pl, b, false)) return rc;
    MDF_REQUIRE(out && which >= 0 && which < (int32_t)e->lms.size(), "engine_lm_features_host: no language model %d", which);
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    MDF_HIP(g.err);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = plan_mirror(pl, e->device, st)) return rc;
    if (int rc = plan_lm_groups(pl, e->cfg.lm_batch, (int64_t)(e->cfg.lm_workspace_gib * 1073741824.0), e->lm_hidden_max, st)) return rc;
    mdf_lm *lm = e->lms[(size_t)which];
    const size_t H = (size_t)mdf_lm_hidden(lm);
    std::vector<float> host;
    mdf_batch_dev view;   // the batch in the plan's order
    if (int rc = batch_in_plan_order(e->perm, &e->generation, pl, b, st, &view)) return rc;
    b = &view;
    // the output is packed in the CALLER's order: position of plan entry p = offset of input protein order[p]
    std::vector<size_t> out_pos((size_t)pl->B + 1, 0);
    {
        std::vector<size_t> len_in((size_t)pl->B);
        for (int32_t p = 0; p < pl->B; ++p) len_in[pl->order.empty() ? (size_t)p : (size_t)pl->order[(size_t)p]] = (size_t)pl->Lq[(size_t)p];
        std::vector<size_t> start_in((size_t)pl->B + 1, 0);
        for (int32_t q = 0; q < pl->B; ++q) start_in[(size_t)q + 1] = start_in[(size_t)q] + len_in[(size_t)q] * H;
        for (int32_t p = 0; p < pl->B; ++p) out_pos[(size_t)p] = start_in[pl->order.empty() ? (size_t)p : (size_t)pl->order[(size_t)p]];
    }
    for (const LmGroup &grp : pl->lm_groups) {
        const int64_t rows = grp.bases.back();
        if (int rc = e->seq_all.grow((size_t)rows, &e->generation)) return rc;
        if (int rc = e->lm_ws.grow(mdf_lm_workspace_bytes(lm, grp.B, grp.Lmax), &e->generation)) return rc;
        if (int rc = e->lm_h[(size_t)which].grow((size_t)rows * H * 4, &e->generation, true)) return rc;
        uint8_t *seq_all = e->seq_all.as<uint8_t>();
        for (int ci = grp.c0; ci < grp.c1; ++ci)
            if (int rc = encode_chunk(e, pl, b, ci, seq_all + grp.bases[(size_t)(ci - grp.c0)], st)) return rc;
        if (int rc = mdf_lm_forward_dev(lm, seq_all, pl->d_lm_rows + grp.rows_pos, pl->d_lm_lens + grp.lens_pos, grp.lens_host.data(), grp.B,
                                        e->lm_h[(size_t)which].as<float>(), e->lm_ws.p, e->lm_ws.bytes, st))
            return rc;
        host.resize((size_t)rows * H);
        MDF_HIP(hipMemcpyAsync(host.data(), e->lm_h[(size_t)which].p, (size_t)rows * H * 4, hipMemcpyDeviceToHost, st));
        MDF_HIP(hipStreamSynchronize(st));
        for (int ci = grp.c0; ci < grp.c1; ++ci) {
            const PlanChunk &ch = pl->chunks[(size_t)ci];
            const int32_t *ro = pl->chunk_row_off.data() + ch.row_off_pos;
            for (int32_t p = ch.p0; p < ch.p1; ++p) {
                const size_t r0 = (size_t)(grp.bases[(size_t)(ci - grp.c0)] + ro[p - ch.p0]);
                memcpy(out + out_pos[(size_t)p], host.data() + r0 * H, (size_t)pl->Lq[(size_t)p] * H * 4);
            }
        }
    }
    return MDF_OK;
}

// ---- everything in one call, host buffers -------------------------------------------------------------------------------
// Round 6: the call is a pipeline of two slots.  mdf_engine_submit_alignments_host packs a batch into the slot's PINNED staging block,
// sends it over on a copy stream, enqueues plan + fused forward on the engine's compute stream behind it and the download into pinned
// memory on a third stream behind that -- and returns; mdf_engine_collect_host waits for the slot's last event, validates the flags
// that came back with the scores (one automatic re-run with a larger CSR capacity) and hands the scores over.  While batch k computes,
// the caller packs batch k + 1 and unpacks batch k - 1: host lists -> host arrays at the device-resident rate (bench.py `binding` leg).
// mdf_engine_run_alignments_host is submit + collect: one code path, the same bits.
static int pinned_grow(char **p, size_t *have, size_t need)
{
    if (*p && *have >= need) return MDF_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr, *have = 0;
    need = align_up(need + need / 8, 4096);
    MDF_HIP(hipHostMalloc(reinterpret_cast<void **>(p), need, hipHostMallocDefault));
    *have = need;
    return MDF_OK;
}

static void host_slots_release(mdf_engine *e)
{
    for (mdf_host_slot &s : e->hslot) {
        if (s.pin_in) (void)hipHostFree(s.pin_in);
        if (s.pin_out) (void)hipHostFree(s.pin_out);
        s.d_in.release(), s.d_out.release();
        if (s.plan) mdf_plan_free(s.plan);
        for (hipEvent_t ev : {s.ev_in, s.ev_comp, s.ev_done})
            if (ev) (void)hipEventDestroy(ev);
        s = mdf_host_slot();
    }
    for (hipStream_t *st : {&e->hs_in, &e->hs_comp, &e->hs_out}) {
        if (*st) (void)hipStreamDestroy(*st);
        *st = nullptr;
    }
}

// the flags of a finished batch (host copies) -> the verdict of mdf_engine_check.  seqs / seq_off / Lq: the batch's own sequences (host), for
// the error path that names the first invalid residue in the CALLER's order
static int verdict_from_flags(const mdf_engine *e, const mdf_plan *pl, const int64_t *bad, const int32_t *status, const char *seqs, const int32_t *seq_off,
                              const int32_t *Lq, int64_t *info)
{
    const size_t nC = pl->chunks.size();
    if (info) info[0] = info[1] = info[2] = info[3] = -1;
    for (size_t ci = 0; ci < nC; ++ci) {
        if (bad[ci] == -1) continue;
        int64_t p = pl->chunks[ci].p0 + (bad[ci] >> 32), pos = bad[ci] & 0xffffffffLL;
        if (!pl->order.empty()) {   // the plan visits the proteins shortest first: the report is the first invalid residue in the caller's order
            bool ok[256] = {false};
            for (const char *a = "-DGULNTKHYWCPVSOIEFXQABZRM"; *a; ++a) ok[(unsigned char)*a] = true;   // the residue alphabet of predict.pyx:26
            p = pl->order[(size_t)p];
            for (int32_t q = 0, found = 0; q < pl->B && !found; ++q)
                for (int32_t i = 0; i < Lq[q]; ++i)
                    if (!ok[(unsigned char)seqs[(size_t)seq_off[q] + (size_t)i]]) {
                        p = q, pos = i, found = 1;
                        break;
                    }
        }
        if (info) info[0] = p, info[1] = pos;
        return fail(MDF_EBADCHAR, "Invalid character in sequence: protein %lld, position %lld", (long long)p, (long long)pos);
    }
    int32_t too_long = 0, need = 0;
    bool overflow = false;
    for (size_t ci = 0; ci < nC; ++ci) {
        too_long = std::max(too_long, status[ci * 4 + 2]);
        if (status[ci * 4]) overflow = true, need = std::max(need, status[ci * 4 + 1]);
    }
    if (too_long) {
        if (info) info[2] = too_long;
        return fail(MDF_EINVAL, "a query of length %d exceeds the max_len the contact stage was given", too_long);
    }
    if (overflow) {
        if (info) info[3] = need;
        return fail(MDF_ECAPACITY, "CSR capacity %lld too small (a chunk needs %d); raise nnz_per_row", (long long)(e ? e->nnz_cap : 0), need);
    }
    return MDF_OK;
}

// plan mirror + fused forward of the slot's batch on the compute stream, then scores + flags into the slot's pinned block on the download stream
static int host_slot_enqueue(mdf_engine *e, mdf_host_slot &s)
{
    char *d = s.d_in.as<char>();
    const size_t nC = s.plan->chunks.size();
    MDF_HIP(hipStreamWaitEvent(e->hs_comp, s.ev_in, 0));
    MDF_HIP(hipMemsetAsync(d + s.o_status, 0, nC * 16, e->hs_comp));
    MDF_HIP(hipMemsetAsync(d + s.o_bad, 0xff, nC * 8, e->hs_comp));
    // a one-shot plan (fresh serial on every call): issued eagerly, never through the graph cache -- its entry could not be
    // seen again and would only push the captured graphs of the serving callers out of the LRU
    if (int rc = ensure(e, s.plan->max_chunk_rows, s.plan->B, s.plan->max_len, s.plan->max_groups)) return rc;
    ++e->eager_runs;
    std::vector<float *> d_scores(e->models.size());
    for (size_t k = 0; k < e->models.size(); ++k) d_scores[k] = reinterpret_cast<float *>(s.d_out.as<char>() + s.s_off[k]);
    if (int rc = forward_alignments_eager(e, s.plan, &s.b, d_scores.data(), nullptr, e->hs_comp)) return rc;
    MDF_HIP(hipEventRecord(s.ev_comp, e->hs_comp));
    MDF_HIP(hipStreamWaitEvent(e->hs_out, s.ev_comp, 0));
    MDF_HIP(hipMemcpyAsync(s.pin_out, s.d_out.p, s.out_flags, hipMemcpyDeviceToHost, e->hs_out));
    MDF_HIP(hipMemcpyAsync(s.pin_out + s.out_flags, d + s.o_status, nC * 16, hipMemcpyDeviceToHost, e->hs_out));
    MDF_HIP(hipMemcpyAsync(s.pin_out + s.out_flags + nC * 16, d + s.o_bad, nC * 8, hipMemcpyDeviceToHost, e->hs_out));
    MDF_HIP(hipEventRecord(s.ev_done, e->hs_out));
    return MDF_OK;
}

extern "C" int mdf_engine_submit_alignments_host(mdf_engine *e, const char *seqs, const int32_t *Lq, int32_t B, const float *coords, const int32_t *Lt,
                                                 const char *q_aln, const char *t_aln, const int32_t *La, int64_t *ticket)
{
    MDF_REQUIRE(e && seqs && Lq && coords && Lt && q_aln && t_aln && La && ticket, "engine_submit_alignments_host: NULL argument");
    MDF_REQUIRE(B > 0, "engine_submit_alignments_host: empty batch");
    *ticket = -1;
    // offsets + the consistency the Python packer checks (a gapped query must spell its sequence)
    std::vector<int32_t> seq_off((size_t)B + 1, 0), coord_off((size_t)B + 1, 0), aln_off((size_t)B + 1, 0);
    int64_t so = 0, co = 0, ao = 0;
    for (int32_t p = 0; p < B; ++p) {
        MDF_REQUIRE(Lq[p] > 0 && Lt[p] >= 0 && La[p] >= 0, "engine_run_alignments_host: bad length at protein %d", p);
        seq_off[(size_t)p] = (int32_t)so, coord_off[(size_t)p] = (int32_t)co, aln_off[(size_t)p] = (int32_t)ao;
        int64_t nongap = 0;
        for (int32_t i = 0; i < La[p]; ++i) nongap += q_aln[ao + i] != '-';
        MDF_REQUIRE(nongap == Lq[p], "protein %d: gapped query does not spell a sequence of length %d", p, Lq[p]);
        so += Lq[p], co += Lt[p], ao += La[p];
        MDF_REQUIRE(so < 0x7fffffffLL && co < 0x7fffffffLL / 3 && ao < 0x7fffffffLL, "batch too large for int32 offsets; split it");
    }
    seq_off[(size_t)B] = (int32_t)so, coord_off[(size_t)B] = (int32_t)co, aln_off[(size_t)B] = (int32_t)ao;
    mdf_plan *pl = nullptr;
    if (int rc = mdf_plan_create(Lq, B, e->cfg.max_rows, e->cfg.max_segment_groups, &pl)) return rc;
    struct PlanGuard {
        mdf_plan *p;
        ~PlanGuard() { if (p) mdf_plan_free(p); }
    } pg{pl};
    const size_t nC = pl->chunks.size();
    std::unique_lock<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    MDF_HIP(g.err);
    mdf_host_slot &s = e->hslot[(size_t)(e->next_ticket % 2)];
    if (s.ticket >= 0)
        return fail(MDF_EINVAL, "engine_submit_alignments_host: two batches are in flight; collect ticket %lld first", (long long)s.ticket);
    if (!e->hs_comp) {
        int lo = 0, hi = 0;
        MDF_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        MDF_HIP(hipStreamCreateWithPriority(&e->hs_in, hipStreamNonBlocking, hi));   // the transfers: queues of their own, under the other batch's kernels
        MDF_HIP(hipStreamCreateWithPriority(&e->hs_out, hipStreamNonBlocking, hi));
        MDF_HIP(hipStreamCreateWithFlags(&e->hs_comp, hipStreamNonBlocking));
    }
    for (hipEvent_t *ev : {&s.ev_in, &s.ev_comp, &s.ev_done})
        if (!*ev) MDF_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    // one block for the inputs, the same layout in pinned and device memory: [seqs | seq_off | Lq | coords | coord_off | q_aln | t_aln | aln_off | status | bad]
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t r = o; o = align_up(o + std::max<size_t>(bytes, 4), 256); return r; };
    const size_t o_seq = take((size_t)so), o_soff = take(((size_t)B + 1) * 4), o_lq = take((size_t)B * 4), o_xyz = take((size_t)co * 12),
                 o_coff = take(((size_t)B + 1) * 4), o_q = take((size_t)ao), o_t = take((size_t)ao), o_aoff = take(((size_t)B + 1) * 4);
    const size_t upload = o;
    const size_t o_status = take(nC * 16), o_bad = take(nC * 8);
    if (int rc = pinned_grow(&s.pin_in, &s.pin_in_bytes, upload)) return rc;
    if (int rc = s.d_in.grow(o, nullptr)) return rc;
    memcpy(s.pin_in + o_seq, seqs, (size_t)so);
    memcpy(s.pin_in + o_soff, seq_off.data(), ((size_t)B + 1) * 4);
    memcpy(s.pin_in + o_lq, Lq, (size_t)B * 4);
    if (co) memcpy(s.pin_in + o_xyz, coords, (size_t)co * 12);
    memcpy(s.pin_in + o_coff, coord_off.data(), ((size_t)B + 1) * 4);
    if (ao) memcpy(s.pin_in + o_q, q_aln, (size_t)ao), memcpy(s.pin_in + o_t, t_aln, (size_t)ao);
    memcpy(s.pin_in + o_aoff, aln_off.data(), ((size_t)B + 1) * 4);
    char *d = s.d_in.as<char>();
    MDF_HIP(hipMemcpyAsync(d, s.pin_in, upload, hipMemcpyHostToDevice, e->hs_in));
    if (int rc = plan_mirror(pl, e->device, e->hs_in)) return rc;   // (its small tables are pageable: staged on the copy stream, not behind the other batch's kernels)
    MDF_HIP(hipEventRecord(s.ev_in, e->hs_in));
    s.b = mdf_batch_dev{};
    s.b.B = B;
    s.b.seqs = d + o_seq;
    s.b.seq_off = reinterpret_cast<const int32_t *>(d + o_soff);
    s.b.Lq = reinterpret_cast<const int32_t *>(d + o_lq);
    s.b.coords = reinterpret_cast<const float *>(d + o_xyz);
    s.b.coord_off = reinterpret_cast<const int32_t *>(d + o_coff);
    s.b.q_aln = d + o_q;
    s.b.t_aln = d + o_t;
    s.b.aln_off = reinterpret_cast<const int32_t *>(d + o_aoff);
    s.b.status = reinterpret_cast<int32_t *>(d + o_status);
    s.b.bad = reinterpret_cast<int64_t *>(d + o_bad);
    s.B = B, s.in_bytes = o, s.o_seq = o_seq, s.o_soff = o_soff, s.o_lq = o_lq, s.o_status = o_status, s.o_bad = o_bad;
    // score blocks of the heads, then the flags: [scores head 0 | ... | status | bad]
    s.s_off.assign(e->models.size(), 0);
    size_t s_total = 0;
    for (size_t k = 0; k < e->models.size(); ++k) {
        s.s_off[k] = s_total;
        s_total = align_up(s_total + (size_t)B * (size_t)mdf_model_num_terms(e->models[k]) * 4, 256);
    }
    s.out_flags = s_total, s.out_bytes = s_total + nC * 24;
    if (int rc = s.d_out.grow(s_total, nullptr)) return rc;
    if (int rc = pinned_grow(&s.pin_out, &s.pin_out_bytes, s.out_bytes)) return rc;
    if (s.plan) mdf_plan_free(s.plan);
    s.plan = pl;
    pg.p = nullptr;   // the slot owns the plan now
    if (int rc = host_slot_enqueue(e, s)) return rc;
    s.ticket = *ticket = e->next_ticket++;
    return MDF_OK;
}

extern "C" int mdf_engine_collect_host(mdf_engine *e, int64_t ticket, float *const *scores_host, int64_t info[4])
{
    MDF_REQUIRE(e && scores_host && ticket >= 0, "engine_collect_host: bad argument");
    std::unique_lock<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    MDF_HIP(g.err);
    mdf_host_slot &s = e->hslot[(size_t)(ticket % 2)];
    MDF_REQUIRE(s.ticket == ticket, "engine_collect_host: ticket %lld is not in flight", (long long)ticket);
    for (size_t k = 0; k < e->models.size(); ++k) MDF_REQUIRE(scores_host[k], "engine_collect_host: scores_host[%zu] is NULL", k);
    struct Release {   // whatever happens below, the slot is free afterwards
        mdf_host_slot &s;
        ~Release() { s.ticket = -1; }
    } rel{s};
    const size_t nC = s.plan->chunks.size();
    int64_t inf[4] = {-1, -1, -1, -1};
    int rc = MDF_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        lk.unlock();   // (the wait does not need the engine: the other slot may be submitted meanwhile)
        const hipError_t werr = hipEventSynchronize(s.ev_done);
        lk.lock();
        MDF_HIP(werr);
        const int32_t *status = reinterpret_cast<const int32_t *>(s.pin_out + s.out_flags);
        const int64_t *bad = reinterpret_cast<const int64_t *>(s.pin_out + s.out_flags + nC * 16);
        rc = verdict_from_flags(e, s.plan, bad, status, s.pin_in + s.o_seq, reinterpret_cast<const int32_t *>(s.pin_in + s.o_soff),
                                reinterpret_cast<const int32_t *>(s.pin_in + s.o_lq), inf);
        if (rc != MDF_ECAPACITY || attempt == 1) break;
        // a denser batch than the CSR arrays planned for: once everything in flight has left the workspaces, raise the capacity and run this
        // batch again from its device copy (the other slot's results are in its own blocks already)
        MDF_HIP(hipStreamSynchronize(e->hs_comp));
        e->cfg.nnz_per_row = (int32_t)(inf[3] / std::max<int64_t>(s.plan->max_chunk_rows, 1) + 8);
        e->rows_alloc = 0;
        MDF_HIP(hipEventRecord(s.ev_in, e->hs_in));
        if ((rc = host_slot_enqueue(e, s))) break;
    }
    if (info) memcpy(info, inf, sizeof(inf));
    if (rc) return rc;
    for (size_t k = 0; k < e->models.size(); ++k)
        memcpy(scores_host[k], s.pin_out + s.s_off[k], (size_t)s.B * (size_t)mdf_model_num_terms(e->models[k]) * 4);
    return MDF_OK;
}

extern "C" int mdf_engine_run_alignments_host(mdf_engine *e, const char *seqs, const int32_t *Lq, int32_t B, const float *coords, const int32_t *Lt,
                                              const char *q_aln, const char *t_aln, const int32_t *La, float *const *scores_host, int64_t info[4])
{
    MDF_REQUIRE(e && seqs && Lq && coords && Lt && q_aln && t_aln && La && scores_host, "engine_run_alignments_host: NULL argument");
    MDF_REQUIRE(B > 0, "engine_run_alignments_host: empty batch");
    int64_t ticket = -1;
    if (int rc = mdf_engine_submit_alignments_host(e, seqs, Lq, B, coords, Lt, q_aln, t_aln, La, &ticket)) return rc;
    return mdf_engine_collect_host(e, ticket, scores_host, info);
}
