/*
 * pnr_oracle_recon.c -- CPU restatement (C99) of the reference's graph post-processing chain
 * reconstruct() (Advantra_plugin.cpp:2096-2181): interpolate_nodelist (:780-861) -> non_blurring
 * mean-shift (:968-1052) -> group1 (:1566-1642) + check_nbr (:1532-1564) -> compute_trees / bfs2
 * (:379-478, :524-530) -> extract_trees (:591-629) -> interpolate_treelist (:714-778).
 *
 * TEST INFRASTRUCTURE ONLY (see pnr_oracle.h).  PARITY UNPINNED: these functions live in the plugin
 * translation unit, which needs Qt4 + the Vaa3D SDK and cannot be built here; restated from the source
 * text with its plain O(n^2) loops and its f32/f64 mixing.  Two notes:
 *   - the shipped v2 source nests `if (!ENFORCE_SINGLE_TREE)` inside `if (ENFORCE_SINGLE_TREE)` (:2142-2166) and
 *     therefore writes no final SWC with default settings; the evident intent (and the commented variant at
 *     :2171) -- extract_trees(TREE_SIZE_MIN) -> interpolate_treelist(1.0, AXON) -> save -- is what is restated.
 *   - group1 sorts node indices by corr with an unstable std::sort (:1570); ties keep index order here, NaN corr sorts last.
 */
#include "pnr_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int *v; int n, cap; } ivec;
static void iv_push(ivec *a, int x)
{
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 4; a->v = (int *)realloc(a->v, sizeof(int) * (size_t)a->cap); }
    a->v[a->n++] = x;
}
static void iv_copy(ivec *d, const ivec *s)
{
    d->n = d->cap = 0; d->v = NULL;
    for (int i = 0; i < s->n; i++) iv_push(d, s->v[i]);
}
static int iv_find(const ivec *a, int x) { for (int i = 0; i < a->n; i++) if (a->v[i] == x) return i; return a->n; }

typedef struct { float x, y, z, vx, vy, vz, corr, sig; int type; ivec nbr; } rnode;
typedef struct { rnode *v; long n, cap; } nvec;
static void nv_push(nvec *a, rnode x)
{
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 64; a->v = (rnode *)realloc(a->v, sizeof(rnode) * (size_t)a->cap); }
    a->v[a->n++] = x;
}
static rnode node_copy(const rnode *s) { rnode d = *s; iv_copy(&d.nbr, &s->nbr); return d; }
static void nv_free(nvec *a) { for (long i = 0; i < a->n; i++) free(a->v[i].nbr.v); free(a->v); a->v = NULL; a->n = a->cap = 0; }
static rnode mk(float x, float y, float z, float vx, float vy, float vz, float corr, float sig, int type)
{
    rnode r; r.x = x; r.y = y; r.z = z; r.vx = vx; r.vy = vy; r.vz = vz; r.corr = corr; r.sig = sig; r.type = type;
    r.nbr.v = NULL; r.nbr.n = r.nbr.cap = 0;
    return r;
}
static float dist3(const rnode *a, const rnode *b) /* sqrt(pow(f,2)+pow(f,2)+pow(f,2)): f64, stored f32 */
{
    double dx = (double)(b->x - a->x), dy = (double)(b->y - a->y), dz = (double)(b->z - a->z);
    return (float)sqrt(dx * dx + dy * dy + dz * dz);
}

/* ---- interpolate_nodelist, bidirectional links (:780-861) ---- */
static void interpolate_nodelist(nvec *nX, float step)
{
    long init = nX->n;
    ivec *chk = (ivec *)calloc((size_t)init, sizeof(ivec));
    for (long i = 0; i < init; i++) for (int j = 0; j < nX->v[i].nbr.n; j++) iv_push(&chk[i], 0);
    for (long i = 1; i < init; ++i)
        for (int j = 0; j < nX->v[i].nbr.n; ++j) {
            if (chk[i].v[j]) continue;
            long i1 = nX->v[i].nbr.v[j];
            int j1 = iv_find(&nX->v[i1].nbr, (int)i);
            if (j1 >= nX->v[i1].nbr.n) continue;
            chk[i].v[j] = 1;
            chk[i1].v[j1] = 1;
            float vnorm = dist3(&nX->v[i], &nX->v[i1]);
            float vx = (nX->v[i1].x - nX->v[i].x) / vnorm, vy = (nX->v[i1].y - nX->v[i].y) / vnorm, vz = (nX->v[i1].z - nX->v[i].z) / vnorm;
            int N = (int)ceilf(vnorm / step);
            for (int k = 1; k < N; ++k) {
                const rnode a = nX->v[i], b = nX->v[i1];
                rnode add = mk(a.x + k * (vnorm / N) * vx, a.y + k * (vnorm / N) * vy, a.z + k * (vnorm / N) * vz, vx, vy, vz,
                               a.corr + (b.corr - a.corr) * (k / (float)N), a.sig + (b.sig - a.sig) * (k / (float)N),
                               (k <= N / 2) ? a.type : b.type);
                nv_push(nX, add);
                long last = nX->n - 1;
                if (k == 1) { iv_push(&nX->v[last].nbr, (int)i); nX->v[i].nbr.v[j] = (int)last; }
                else { iv_push(&nX->v[last].nbr, (int)(last - 1)); iv_push(&nX->v[last - 1].nbr, (int)last); }
                if (k == N - 1) { iv_push(&nX->v[last].nbr, (int)i1); nX->v[i1].nbr.v[j1] = (int)last; }
            }
        }
    for (long i = 0; i < init; i++) free(chk[i].v);
    free(chk);
}

/* ---- non_blurring mean-shift (:968-1052); nY = nX with refined x,y,z,sig ---- */
static void non_blurring(const nvec *nX, nvec *nY, float SIG2RAD, int MAXITER, float EPS2)
{
    for (long i = 0; i < nX->n; i++) nv_push(nY, node_copy(&nX->v[i]));
    for (long i = 1; i < nY->n; ++i) {
        float conv[4] = {nX->v[i].x, nX->v[i].y, nX->v[i].z, nX->v[i].sig}, next[4];
        int iter = 0, cnt;
        float d2;
        do {
            cnt = 0;
            next[0] = next[1] = next[2] = next[3] = 0;
            float r2 = (float)pow((double)(SIG2RAD * conv[3]), 2);
            for (long j = 1; j < nX->n; ++j) {
                float x2 = (float)pow((double)(nX->v[j].x - conv[0]), 2);
                if (x2 <= r2) {
                    float y2 = (float)pow((double)(nX->v[j].y - conv[1]), 2);
                    if (x2 + y2 <= r2) {
                        float z2 = (float)pow((double)(nX->v[j].z - conv[2]), 2);
                        if (x2 + y2 + z2 <= r2) {
                            next[0] += nX->v[j].x; next[1] += nX->v[j].y; next[2] += nX->v[j].z; next[3] += nX->v[j].sig;
                            cnt++;
                        }
                    }
                }
            }
            next[0] /= cnt; next[1] /= cnt; next[2] /= cnt; next[3] /= cnt;
            d2 = (float)(pow((double)(next[0] - conv[0]), 2) + pow((double)(next[1] - conv[1]), 2) + pow((double)(next[2] - conv[2]), 2));
            conv[0] = next[0]; conv[1] = next[1]; conv[2] = next[2]; conv[3] = next[3];
            iter++;
        } while (iter < MAXITER && d2 > EPS2);
        nY->v[i].x = conv[0]; nY->v[i].y = conv[1]; nY->v[i].z = conv[2]; nY->v[i].sig = conv[3];
    }
}

/* ---- check_nbr (:1532-1564) ---- */
static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }
static void check_nbr(nvec *nX)
{
    for (long i = 1; i < nX->n; ++i) {
        ivec *nb = &nX->v[i].nbr;
        qsort(nb->v, (size_t)nb->n, sizeof(int), cmp_int);
        int m = 0;
        for (int k = 0; k < nb->n; k++) if (k == 0 || nb->v[k] != nb->v[k - 1]) nb->v[m++] = nb->v[k];
        nb->n = m;
        int pos = iv_find(nb, (int)i);
        if (pos < nb->n) { memmove(nb->v + pos, nb->v + pos + 1, sizeof(int) * (size_t)(nb->n - pos - 1)); nb->n--; }
    }
    for (long i = 1; i < nX->n; ++i)
        for (int j = 0; j < nX->v[i].nbr.n; ++j) {
            int o = nX->v[i].nbr.v[j];
            if (iv_find(&nX->v[o].nbr, (int)i) >= nX->v[o].nbr.n) iv_push(&nX->v[o].nbr, (int)i);
        }
}

/* ---- group1, sphere grouping (:1566-1642) ---- */
typedef struct { float corr; long idx; } cidx;
static int cmp_corr_desc(const void *a, const void *b)
{
    const cidx *x = (const cidx *)a, *y = (const cidx *)b;
    /* a NaN corr makes the reference's comparator (:1571) no strict weak order (undefined behaviour in std::sort); the restatement
     * fixes a total order: numbers by decreasing corr, NaNs behind all of them, equal keys by index */
    const int xn = x->corr != x->corr, yn = y->corr != y->corr;
    if (xn != yn) return xn - yn;
    if (!xn) {
        if (x->corr > y->corr) return -1;
        if (x->corr < y->corr) return 1;
    }
    return (x->idx > y->idx) - (x->idx < y->idx);
}
static void group1(nvec *nX, nvec *nY, float rad)
{
    long n = nX->n;
    nX->v[0].corr = FLT_MAX;
    cidx *ord = (cidx *)malloc(sizeof(cidx) * (size_t)n);
    for (long i = 0; i < n; i++) { ord[i].corr = nX->v[i].corr; ord[i].idx = i; }
    qsort(ord, (size_t)n, sizeof(cidx), cmp_corr_desc);
    long *X2Y = (long *)malloc(sizeof(long) * (size_t)n);
    for (long i = 0; i < n; i++) X2Y[i] = -1;
    X2Y[0] = 0;
    nv_push(nY, node_copy(&nX->v[0]));
    for (long i = 1; i < n; ++i) /* soma nodes as independent groups at the beginning (:1580-1588) */
        if (nX->v[i].type == 1) {
            X2Y[i] = nY->n;
            nv_push(nY, node_copy(&nX->v[i]));
        }
    for (long i = 1; i < n; ++i) {
        long ci = ord[i].idx;
        if (X2Y[ci] != -1) continue;
        X2Y[ci] = nY->n;
        rnode g = node_copy(&nX->v[ci]);
        float grp = 1;
        float r2 = rad * rad;
        for (long j = 1; j < n; ++j) {
            if (j != ci && X2Y[j] == -1) {
                float d2 = (float)pow((double)(nX->v[j].x - nX->v[ci].x), 2);
                if (d2 <= r2) {
                    d2 = (float)(d2 + pow((double)(nX->v[j].y - nX->v[ci].y), 2));
                    if (d2 <= r2) {
                        d2 = (float)(d2 + pow((double)(nX->v[j].z - nX->v[ci].z), 2));
                        if (d2 <= r2) {
                            X2Y[j] = nY->n;
                            for (int k = 0; k < nX->v[j].nbr.n; ++k) iv_push(&g.nbr, nX->v[j].nbr.v[k]);
                            grp++;
                            float a = (grp - 1) / grp;
                            float b = (float)(1.0 / grp);
                            g.x = a * g.x + b * nX->v[j].x;
                            g.y = a * g.y + b * nX->v[j].y;
                            g.z = a * g.z + b * nX->v[j].z;
                            g.sig = a * g.sig + b * nX->v[j].sig;
                            g.corr = a * g.corr + b * nX->v[j].corr;
                        }
                    }
                }
            }
        }
        g.type = 2; /* AXON */
        nv_push(nY, g);
    }
    for (long i = 1; i < nY->n; ++i)
        for (int j = 0; j < nY->v[i].nbr.n; ++j) nY->v[i].nbr.v[j] = (int)X2Y[nY->v[i].nbr.v[j]];
    check_nbr(nY);
    free(ord);
    free(X2Y);
}

/* ---- bfs2 / compute_trees (:379-478) ---- */
static void bfs2(const nvec *nl, nvec *tree, int remove_isolated)
{
    long n = nl->n;
    int *dist = (int *)malloc(sizeof(int) * (size_t)n), *nmap = (int *)malloc(sizeof(int) * (size_t)n), *parent = (int *)malloc(sizeof(int) * (size_t)n);
    int *queue = (int *)malloc(sizeof(int) * (size_t)n);
    for (long i = 0; i < n; i++) { dist[i] = 2147483647; nmap[i] = -1; parent[i] = -1; }
    dist[0] = -1;
    nv_push(tree, node_copy(&nl->v[0]));
    int treecnt = 0;
    for (;;) {
        long seed = -1;
        for (long i = 1; i < n; i++) if (dist[i] == 2147483647) { seed = i; break; }
        if (seed <= 0) break;
        treecnt++;
        dist[seed] = 0; nmap[seed] = -1; parent[seed] = -1;
        long qh = 0, qt = 0;
        queue[qt++] = (int)seed;
        int nodesInTree = 0;
        while (qh < qt) {
            int curr = queue[qh++];
            rnode t = nl->v[curr];
            t.nbr.v = NULL; t.nbr.n = t.nbr.cap = 0;
            if (t.type != 1) t.type = treecnt + 2;
            if (parent[curr] > 0) iv_push(&t.nbr, nmap[parent[curr]]);
            nmap[curr] = (int)tree->n;
            nv_push(tree, t);
            nodesInTree++;
            for (int j = 0; j < nl->v[curr].nbr.n; j++) {
                int adj = nl->v[curr].nbr.v[j];
                if (dist[adj] == 2147483647) { dist[adj] = dist[curr] + 1; parent[adj] = curr; queue[qt++] = adj; }
            }
            if (nodesInTree == 1 && qh == qt && remove_isolated) {
                free(tree->v[tree->n - 1].nbr.v);
                tree->n--;
                nmap[curr] = -1;
            }
        }
    }
    free(dist); free(nmap); free(parent); free(queue);
}

/* ---- extract_trees (:591-629) ---- */
static void extract_trees(const nvec *X, nvec *Y, int min_size)
{
    long n = X->n;
    char *rm = (char *)calloc((size_t)n + 1, 1);
    long root_curr = 1, root_prev = 1;
    for (long i = 1; i <= n; ++i)
        if (i == n || X->v[i].nbr.n == 0) {
            root_prev = root_curr;
            root_curr = i;
            if (root_curr - root_prev < min_size) for (long j = root_prev; j < root_curr; ++j) rm[j] = 1;
        }
    long *X2Y = (long *)malloc(sizeof(long) * (size_t)n);
    for (long i = 0; i < n; ++i) {
        X2Y[i] = -1;
        if (!rm[i]) { X2Y[i] = Y->n; nv_push(Y, node_copy(&X->v[i])); }
    }
    for (long i = 1; i < Y->n; ++i) for (int j = 0; j < Y->v[i].nbr.n; ++j) Y->v[i].nbr.v[j] = (int)X2Y[Y->v[i].nbr.v[j]];
    free(rm); free(X2Y);
}

/* ---- extract_largest_tree (:546-589), the ENFORCE_SINGLE_TREE branch of reconstruct() (:2142-2152) ---- */
static void extract_largest_tree(const nvec *X, nvec *Y)
{
    long n = X->n;
    long root_curr = 1, root_prev = 1, tree_max_size = -2147483647L, tree_max_beg = -2147483647L, tree_max_end = -2147483647L;
    for (long i = 1; i <= n; ++i)
        if (i == n || X->v[i].nbr.n == 0) {
            root_prev = root_curr;
            root_curr = i;
            if (root_curr - root_prev > tree_max_size) { tree_max_size = root_curr - root_prev; tree_max_beg = root_prev; tree_max_end = root_curr; }
        }
    long *X2Y = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));
    for (long i = 0; i < n; ++i) {
        X2Y[i] = -1;
        if (i == 0 || (i >= tree_max_beg && i < tree_max_end)) { X2Y[i] = Y->n; nv_push(Y, node_copy(&X->v[i])); }
    }
    for (long i = 1; i < Y->n; ++i) for (int j = 0; j < Y->v[i].nbr.n; ++j) Y->v[i].nbr.v[j] = (int)X2Y[Y->v[i].nbr.v[j]];
    free(X2Y);
}

/* ---- interpolate_treelist, one-directional links (:714-778) ---- */
static void interpolate_treelist(nvec *t, float step, int type)
{
    long init = t->n;
    for (long i = 1; i < init; ++i) {
        if (type >= 0 && t->v[i].type != 1) t->v[i].type = type;
        for (int j = 0; j < t->v[i].nbr.n; ++j) {
            long i1 = t->v[i].nbr.v[j];
            float vnorm = dist3(&t->v[i], &t->v[i1]);
            float vx = (t->v[i1].x - t->v[i].x) / vnorm, vy = (t->v[i1].y - t->v[i].y) / vnorm, vz = (t->v[i1].z - t->v[i].z) / vnorm;
            int N = (int)ceilf(vnorm / step);
            for (int k = 1; k < N; ++k) {
                const rnode a = t->v[i], b = t->v[i1];
                rnode add = mk(a.x + k * (vnorm / N) * vx, a.y + k * (vnorm / N) * vy, a.z + k * (vnorm / N) * vz, vx, vy, vz,
                               a.corr + (b.corr - a.corr) * (k / (float)N), a.sig + (b.sig - a.sig) * (k / (float)N),
                               (k <= N / 2) ? a.type : b.type);
                nv_push(t, add);
                long last = t->n - 1;
                if (k == 1) t->v[i].nbr.v[j] = (int)last;
                else iv_push(&t->v[last - 1].nbr, (int)last);
                if (k == N - 1) iv_push(&t->v[last].nbr, (int)i1);
            }
        }
    }
}

/* reconstruct(n0): nodes in (incl. dummy 0) + link pairs -> final tree list: out nodes + parent (-1 = root).
 * returns the number of output nodes (incl. dummy 0); writes at most cap. */
int64_t orc_reconstruct(const orc_node *nodes, int64_t n_nodes, const int32_t *links, int64_t n_links, float trace_rsmpl,
                        float sig2radius, int refine_iter, float epsilon2, float group_radius, int tree_size_min,
                        orc_node *out, int32_t *parent, int64_t cap)
{
    nvec n0 = {0, 0, 0}, n1 = {0, 0, 0}, n2 = {0, 0, 0}, tr = {0, 0, 0}, t3 = {0, 0, 0};
    for (int64_t i = 0; i < n_nodes; i++)
        nv_push(&n0, mk(nodes[i].x, nodes[i].y, nodes[i].z, nodes[i].vx, nodes[i].vy, nodes[i].vz, nodes[i].corr, nodes[i].sig, nodes[i].type));
    for (int64_t k = 0; k < n_links; k++) { iv_push(&n0.v[links[2 * k]].nbr, links[2 * k + 1]); iv_push(&n0.v[links[2 * k + 1]].nbr, links[2 * k]); }
    interpolate_nodelist(&n0, trace_rsmpl);
    non_blurring(&n0, &n1, sig2radius, refine_iter, epsilon2);
    group1(&n1, &n2, group_radius);
    bfs2(&n2, &tr, 1);
    if (tree_size_min < 0) extract_largest_tree(&tr, &t3); /* ENFORCE_SINGLE_TREE (:2142-2152) */
    else extract_trees(&tr, &t3, tree_size_min);
    interpolate_treelist(&t3, 1.0f, 2 /* AXON */);
    int64_t n = t3.n;
    for (int64_t i = 0; i < n && i < cap; i++) {
        const rnode *r = &t3.v[i];
        out[i].x = r->x; out[i].y = r->y; out[i].z = r->z; out[i].vx = r->vx; out[i].vy = r->vy; out[i].vz = r->vz;
        out[i].corr = r->corr; out[i].sig = r->sig; out[i].type = r->type;
        parent[i] = (i > 0 && r->nbr.n > 0) ? r->nbr.v[0] : -1;
    }
    nv_free(&n0); nv_free(&n1); nv_free(&n2); nv_free(&tr); nv_free(&t3);
    return n;
}
