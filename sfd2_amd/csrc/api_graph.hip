// libsfd2hip: one query unit (extract + k matches) and its hipGraph cache.
#include "sfd2_ctx.h"

// ------------------------------------------------------------------------------------------ extract + match, hipGraph cache
// One query unit of the localisation pipeline: extract one image, match its descriptors against k resident
// database sets.  With option "graphs" the stream work of one unit is captured once per geometry and replayed
// (BASELINE configs[4]: "per-GPU hipGraph capture").  A graph holds raw pointers, so the cache key is every argument
// that ends up in a kernel parameter; entries die when any workspace buffer is reallocated (g_alloc_gen).
struct GraphKey {
    int H, W, top_k, flags, k, dim, n0;
    float conf_th;
    sfd2_match_conf conf;
    const void *img, *kp, *sc, *de, *m, *ms;
    unsigned long long db_hash;
    bool operator==(const GraphKey &o) const { return memcmp(this, &o, sizeof(GraphKey)) == 0; }
};
struct GraphEntry {
    GraphKey key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    unsigned long long gen = 0, last_use = 0;
    int seen = 0;                       // eager passes made with this key (the first one sizes the workspace)
    void *pin = nullptr;                // this entry's own job descriptors: replay re-reads them from pinned memory
    size_t pin_cap = 0;
    DevBuf jobs, fins;
};
#define SFD2_MAX_GRAPHS 16

static void graph_entry_release(GraphEntry &e)
{
    if (e.exec) (void)hipGraphExecDestroy(e.exec);
    if (e.graph) (void)hipGraphDestroy(e.graph);
    if (e.pin) (void)hipHostFree(e.pin);
    e.jobs.release();
    e.fins.release();
    e = GraphEntry();
}

void graphs_release(sfd2_ctx *c)
{
    if (!c->graphs) return;
    for (int i = 0; i < c->n_graphs; ++i) graph_entry_release(c->graphs[i]);
    delete[] c->graphs;
    c->graphs = nullptr;
    c->n_graphs = 0;
}

static int extract_match_eager(sfd2_ctx *c, const void *img, int H, int W, float conf_th, int top_k, int flags, float *kp,
                               float *sc, float *de, const sfd2_desc_set *db, int k, int dim, const sfd2_match_conf *conf,
                               int64_t *m, float *ms)
{
    int n_dummy = 0;
    if (sfd2_extract(c, img, 1, H, W, conf_th, top_k, flags | SFD2_FLAG_ASYNC, kp, sc, de, 1, top_k, &n_dummy)) return -1;
    if (k > 0) {
        const sfd2_desc_set q = {de, top_k, SFD2_DT_F32, SFD2_LAYOUT_ND, 1, nullptr, 0, 0};
        if (sfd2_match_batch(c, &q, db, k, dim, conf, m, ms, 1, SFD2_FLAG_ASYNC)) return -1;
    }
    return 0;
}

extern "C" int sfd2_extract_match(sfd2_ctx *c, const void *img_dev, int H, int W, float conf_th, int top_k, int flags,
                                  float *kpts_xy, float *scores, float *desc, const sfd2_desc_set *db, int k, int dim,
                                  const sfd2_match_conf *conf, int64_t *matches0, float *scores0)
{
    if (!c || !img_dev || !kpts_xy || !scores || !desc) return fail("sfd2_extract_match: null argument");
    if (top_k <= 0) return fail("sfd2_extract_match: top_k must be positive (fixed-capacity device outputs)");
    if (flags & SFD2_FLAG_DESC_STORE64) return fail("sfd2_extract_match: SFD2_FLAG_DESC_STORE64 is for stores; the matcher reads the float [n][128] descriptors");
    if (k < 0 || (k > 0 && (!db || !conf || !matches0 || !scores0))) return fail("sfd2_extract_match: null matcher argument");
    for (int i = 0; i < k; ++i)
        if (!db[i].on_device || db[i].rows) return fail("sfd2_extract_match: database sets must be device resident, without row selection");
    HIPCHECK(hipSetDevice(c->device));
    if (!c->use_graphs || c->prof_max_steps > 0)   // per-launch events cannot be read back from inside a graph
        return extract_match_eager(c, img_dev, H, W, conf_th, top_k, flags, kpts_xy, scores, desc, db, k, dim, conf, matches0, scores0);

    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.H = H; key.W = W; key.top_k = top_k; key.flags = flags; key.k = k; key.dim = dim; key.n0 = top_k; key.conf_th = conf_th;
    if (conf) key.conf = *conf;
    key.img = img_dev; key.kp = kpts_xy; key.sc = scores; key.de = desc; key.m = matches0; key.ms = scores0;
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < k; ++i) {
        const unsigned long long v[3] = {(unsigned long long)(uintptr_t)db[i].data, (unsigned long long)db[i].n,
                                         ((unsigned long long)db[i].dtype << 8) | (unsigned long long)db[i].layout};
        for (unsigned long long x : v) { h ^= x; h *= 1099511628211ull; }
    }
    key.db_hash = h;
    if (!c->graphs) { c->graphs = new GraphEntry[SFD2_MAX_GRAPHS]; c->n_graphs = SFD2_MAX_GRAPHS; }
    GraphEntry *e = nullptr, *lru = &c->graphs[0];
    for (int i = 0; i < c->n_graphs; ++i) {
        GraphEntry &g = c->graphs[i];
        if (g.seen && g.key == key) { e = &g; break; }
        if (g.last_use < lru->last_use) lru = &g;
    }
    if (!e) {   // new geometry: recycle the least recently used slot, run eagerly once (allocations happen here)
        graph_entry_release(*lru);
        e = lru;
        e->key = key;
    }
    e->last_use = ++c->graph_clock;
    if (e->exec && e->gen == g_alloc_gen) {
        HIPCHECK(hipGraphLaunch(e->exec, c->stream));
        return 0;
    }
    if (e->exec) {   // a workspace buffer moved since the capture: the graph's pointers are stale
        (void)hipGraphExecDestroy(e->exec); e->exec = nullptr;
        (void)hipGraphDestroy(e->graph); e->graph = nullptr;
        e->seen = 0;
    }
    if (e->seen == 0) {
        e->seen = 1;
        return extract_match_eager(c, img_dev, H, W, conf_th, top_k, flags, kpts_xy, scores, desc, db, k, dim, conf, matches0, scores0);
    }
    // second sight of the key: capture.  The matcher's job descriptors are copied from pinned host memory by a graph
    // node at every replay, so the entry gets its own pinned block and device copies that no other call rewrites.
    const size_t jb = 2 * (size_t)std::max(k, 1) * sizeof(MatchJob), fb = (size_t)std::max(k, 1) * sizeof(MatchFinal);
    if (!e->pin) {
        HIPCHECK(hipHostMalloc(&e->pin, jb + fb, hipHostMallocDefault));
        e->pin_cap = jb + fb;
        HIPCHECK(e->jobs.ensure(jb + fb));
    }
    HIPCHECK(hipStreamSynchronize(c->stream));
    HIPCHECK(hipEventSynchronize(c->ev_jobs));
    const unsigned long long gen0 = g_alloc_gen.load();
    std::swap(c->pin_jobs, e->pin); std::swap(c->pin_cap, e->pin_cap);
    std::swap(c->m_jobs, e->jobs); std::swap(c->m_fins, e->fins);
    hipError_t be = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
    int rc = -1;
    hipGraph_t g = nullptr;
    if (be == hipSuccess) {
        rc = extract_match_eager(c, img_dev, H, W, conf_th, top_k, flags, kpts_xy, scores, desc, db, k, dim, conf, matches0, scores0);
        const hipError_t ee = hipStreamEndCapture(c->stream, &g);
        if (ee != hipSuccess) rc = fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(ee));
    } else {
        fail(std::string("hipStreamBeginCapture: ") + hipGetErrorString(be));
    }
    std::swap(c->pin_jobs, e->pin); std::swap(c->pin_cap, e->pin_cap);
    std::swap(c->m_jobs, e->jobs); std::swap(c->m_fins, e->fins);
    if (rc != 0 || g_alloc_gen != gen0) {   // an allocation inside the capture means the warm-up pass did not cover it
        if (g) (void)hipGraphDestroy(g);
        e->seen = 0;
        if (rc == 0) return fail("sfd2_extract_match: workspace changed during capture");
        return -1;
    }
    e->graph = g;
    HIPCHECK(hipGraphInstantiate(&e->exec, g, nullptr, nullptr, 0));
    e->gen = g_alloc_gen;
    HIPCHECK(hipGraphLaunch(e->exec, c->stream));
    return 0;
}
