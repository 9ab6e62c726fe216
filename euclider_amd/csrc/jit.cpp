/*
 * jit.cpp -- generator of the scene-specialised trace kernels + hiprtc build + code-object cache (see jit.hpp).
 * Host code only; nothing here traces rays.  The generator reads the SAME flat scene the interpreter kernels read
 * (flat_scene.h), so what it emits is by construction the program the interpreter would have walked.
 */
#include "jit.hpp"

#include <hip/hiprtc.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

namespace euclider {

#define EU_JIT_VERSION "eu-jit-6"

/* the device headers, embedded at build time (csrc/Makefile: jit_headers.inc) */
struct EmbeddedHeader { const char *name; const char *text; };
static const EmbeddedHeader kHeaders[] = {
#include "jit_headers.inc"
};
static constexpr int kNumHeaders = (int)(sizeof(kHeaders) / sizeof(kHeaders[0]));

static const char *const kCompileFlags[] = {
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-c++11-narrowing",
    /* MachineLICM hoists constant materialisation and invariant scalar loads out of the ray loops and lets them live (and spill) across
     * the whole kernel: 3d_room shade 159 -> 0 SGPR spills, 154 -> 125 VGPRs without it */
    "-mllvm", "-disable-machine-licm",
#ifdef EU_LOW_PRECISION
    "-DEU_LOW_PRECISION",
#endif
};

namespace {

struct Out {
    std::string s;
    void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        int n = vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        if (n < 0) return;
        if ((size_t)n < sizeof buf) { s.append(buf, (size_t)n); return; }
        std::string big((size_t)n + 1, '\0');
        va_start(ap, fmt);
        vsnprintf(&big[0], big.size(), fmt, ap);
        va_end(ap);
        s.append(big.c_str(), (size_t)n);
    }
};

/* exact literals: hexadecimal floating point for finite values, bit patterns for the rest */
static std::string lit64(eu_f64 v) {
    char b[64];
    if (std::isfinite(v)) snprintf(b, sizeof b, "%a", v);
    else { uint64_t u; memcpy(&u, &v, 8); snprintf(b, sizeof b, "__builtin_bit_cast(double, 0x%016llxull)", (unsigned long long)u); }
    return b;
}
static std::string litr(real v) {
#if EU_REAL_BITS == 32
    char b[64];
    if (std::isfinite(v)) snprintf(b, sizeof b, "%af", (double)v);
    else { uint32_t u; memcpy(&u, &v, 4); snprintf(b, sizeof b, "__builtin_bit_cast(float, 0x%08xu)", u); }
    return b;
#else
    return lit64(v);
#endif
}

struct OpView { uint32_t kind, count, first, param; };
struct EntityView { uint32_t shape_first, shape_root, material; int32_t surface; uint32_t max_hits, bound; };

struct Gen {
    const uint64_t *w;
    const EuFlatHeader *h;
    int D;
    Out o;
    std::set<std::pair<uint32_t, uint32_t>> inside_done;
    std::string inside_defs;      /* static member functions in_<first>_<root>(p) */

    /* Straight-line code is worth its compile time up to a point (jit.hpp: kJitOpsBudget, kJitSurfacesBudget).  Entities are taken in
     * order while their shape programs fit the budget; the others are traced by the interpreter's routines from the flat scene
     * (trace_device.h: interp_entities, inside_subtree, hit_normal) -- at their place in the entity order.  Surfaces beyond theirs are
     * evaluated from their records (surface_color; its operand stack lies in dynamic LDS: JitPlan::color_stack). */
    uint32_t ops_budget = kJitOpsBudget, surfaces_budget = kJitSurfacesBudget;
    std::vector<uint8_t> straight;      /* per entity */
    uint32_t n_straight_ops = 0, n_interp_entities = 0, n_generic_surfaces = 0;

    explicit Gen(const FlatScene &flat) : w(flat.words.data()), h(&flat.header()), D((int)flat.header().dim) {}

    OpView op(uint32_t i) const {
        const uint64_t x = w[h->off_ops + i];
        return OpView{(uint32_t)(x & 0xff), (uint32_t)((x >> 8) & 0xff), (uint32_t)((x >> 16) & 0xffff), (uint32_t)(x >> 32)};
    }
    EntityView entity(uint32_t e) const {
        const uint64_t a = w[h->off_entities + 2 * e], b = w[h->off_entities + 2 * e + 1];
        return EntityView{(uint32_t)(a & 0xffffu), (uint32_t)((a >> 16) & 0xffffu), (uint32_t)((a >> 32) & 0xffffu), (int32_t)(int16_t)(uint16_t)(a >> 48),
                          (uint32_t)b, (uint32_t)(b >> 32)};
    }
    const real *params(uint32_t off) const { return (const real *)(w + h->off_params) + off; }
    const real *bounds(uint32_t b) const { return (const real *)(w + h->off_bounds) + (uint32_t)(D + 2) * b; }
    const EuFlatSurface *surface(uint32_t s) const { return (const EuFlatSurface *)(w + h->off_surfaces + 8 * s); }
    const EuFlatColorOp *color_op(uint32_t c) const { return (const EuFlatColorOp *)(w + h->off_color_ops + 16 * c); }
    const EuFlatMapped *mapped(uint32_t m) const { return (const EuFlatMapped *)(w + h->off_mapped + 8 * m); }

    static const char *kind_name(uint32_t k) {
        switch (k) {
        case EU_SH_VOID: return "EU_SH_VOID"; case EU_SH_SPHERE: return "EU_SH_SPHERE"; case EU_SH_PLANE: return "EU_SH_PLANE";
        case EU_SH_HALFSPACE: return "EU_SH_HALFSPACE"; case EU_SH_CYLINDER: return "EU_SH_CYLINDER";
        case EU_SH_UNION: return "EU_SH_UNION"; case EU_SH_INTERSECTION: return "EU_SH_INTERSECTION";
        case EU_SH_COMPLEMENT: return "EU_SH_COMPLEMENT"; case EU_SH_SYMDIFF: return "EU_SH_SYMDIFF";
        case EU_SH_CHAIN_UNION: return "EU_SH_CHAIN_UNION"; case EU_SH_CHAIN_INTERSECTION: return "EU_SH_CHAIN_INTERSECTION";
        case EU_SH_CHAIN_BOX: return "EU_SH_CHAIN_BOX"; case EU_SH_CHAIN_BOX0: return "EU_SH_CHAIN_BOX0";
        default: return "EU_SH_VOID";
        }
    }
    uint32_t param_count(const OpView &p) const {      /* reals behind op.param */
        const uint32_t d = (uint32_t)D;
        switch (p.kind) {
        case EU_SH_SPHERE: return d + 2;
        case EU_SH_PLANE: return d + 1;
        case EU_SH_HALFSPACE: return 2 * d + 2;
        case EU_SH_CYLINDER: return 2 * d + 2;
        case EU_SH_CHAIN_UNION: case EU_SH_CHAIN_INTERSECTION: case EU_SH_CHAIN_BOX: case EU_SH_CHAIN_BOX0:
            return p.count * (2 * d + 2) + d + 2;      /* the leaves, then the chain's bounding sphere */
        default: return 0;
        }
    }
    /* `static constexpr real NAME[] = {...};` */
    std::string real_array(const std::string &name, const real *v, uint32_t n) const {
        std::string s = "static constexpr real " + name + "[] = {";
        if (n == 0) s += "R(0.0)";
        for (uint32_t k = 0; k < n; k++) { if (k) s += ", "; s += litr(v[k]); }
        s += "};";
        return s;
    }
    /* Shape parameters and bounding spheres are NOT emitted as constants: they are read from the resident scene at constant offsets
     * (wave-uniform addresses: scalar loads with immediate offsets).  As literals every double costs two s_mov_b32 and the optimiser,
     * seeing through them, reshapes the chain matrices into hundreds of simultaneously live lane masks (measured: 166 VGPRs + 115
     * spilled, 533 SGPR spills for 3d_room's intersect kernel against 105 / 47 with the parameters in memory).  The same holds for
     * the few numbers the closed-form box routines need (chain_slab, chain_inside_box), tried on their own in round 3: with them
     * visible the optimiser shares plane hits between 4d_frame's concentric boxes and keeps them all alive -- 912 SGPR spills,
     * 2.6 against 3.9 Gray/s. */
    bool shape_params_in_memory = true;
    /* (`po`, `bo`: offsets in reals of the entity at hand behind the entity the code was written for -- zero, a constant, except in
     * the loop over a run of congruent entities: find_runs) */
    std::string op_params(const std::string &name, uint32_t i) const {
        const OpView p = op(i);
        if (!shape_params_in_memory) return real_array(name, params(p.param), param_count(p));
        char b[160];
        snprintf(b, sizeof b, "const real *const %s = (const real *)(S.w + %uu) + (%uu + po);", name.c_str(), h->off_params, p.param);
        return b;
    }
    std::string bound_array(const std::string &name, uint32_t bi) const {
        if (!shape_params_in_memory) return real_array(name, bounds(bi), (uint32_t)D + 2);
        char b[160];
        snprintf(b, sizeof b, "const real *const %s = (const real *)(S.w + %uu) + (%uu + bo);", name.c_str(), h->off_bounds, (uint32_t)(D + 2) * bi);
        return b;
    }

    /* Runs of CONGRUENT entities: consecutive surfaced entities whose shape programs are the same operations on parameters and bounding
     * spheres that lie a constant stride apart in the flat scene (a row of identical columns, 4d_cylinders' eight crosses).  Such a run
     * gets ONE body in a loop over its entities -- in entity order, as trace_closest wants -- with the stride added to every address
     * (wave-uniform: still scalar loads): an eighth of the code to compile and to fetch. */
    struct Run { uint32_t e0, count, n_ops, dp, db; };      /* dp: reals between two entities' parameters; db: bounding spheres between theirs */
    std::vector<Run> runs;
    std::vector<int> run_of;           /* per entity: index into runs, or -1 */
    bool congruent(uint32_t ea, uint32_t eb, uint32_t &dp, uint32_t &db, bool &have_dp, bool &have_db) const {
        const EntityView A = entity(ea), B = entity(eb);
        if (A.surface < 0 || B.surface < 0) return false;
        const uint32_t n = A.shape_root - A.shape_first + 1u;
        if (B.shape_root - B.shape_first + 1u != n || B.shape_first != A.shape_first + n) return false;
        if ((A.bound == 0xffffffffu) != (B.bound == 0xffffffffu)) return false;
        auto same = [](uint32_t d, uint32_t &acc, bool &have) { if (!have) { acc = d; have = true; return true; } return acc == d; };
        if (A.bound != 0xffffffffu && (B.bound <= A.bound || !same(B.bound - A.bound, db, have_db))) return false;
        for (uint32_t k = 0; k < n; k++) {
            const OpView p = op(A.shape_first + k), q = op(B.shape_first + k);
            if (p.kind != q.kind || p.count != q.count || p.first - A.shape_first != q.first - B.shape_first) return false;
            if (p.kind == EU_SH_SKIP) { if (q.param <= p.param || !same(q.param - p.param, db, have_db)) return false; }
            else if (p.kind < EU_SH_UNION || p.kind >= EU_SH_CHAIN_UNION) { if (p.kind != EU_SH_VOID && (q.param <= p.param || !same(q.param - p.param, dp, have_dp))) return false; }
        }
        return true;
    }
    void find_runs() {
        const uint32_t ne = h->n_entities;
        run_of.assign(ne, -1);
        if (!shape_params_in_memory || no_runs) return;
        for (uint32_t e = 0; e + 1 < ne;) {
            uint32_t dp = 0, db = 0; bool have_dp = false, have_db = false;
            uint32_t last = e;
            while (last + 1 < ne && congruent(last, last + 1, dp, db, have_dp, have_db)) last++;
            const EntityView E = entity(e);
            const uint32_t n = E.shape_root - E.shape_first + 1u;
            if (last > e && n >= 4u) {      /* (a run of bare leaves is not worth a loop) */
                for (uint32_t k = e; k <= last; k++) run_of[k] = (int)runs.size();
                runs.push_back(Run{e, last - e + 1u, n, have_dp ? dp : 0u, have_db ? db : 0u});
                e = last + 1;
            } else e++;
        }
    }
    bool no_runs = false;              /* tuning / tests: -DEU_JIT_NO_RUNS */

    /* ---- is_point_inside of the subtree ops[first..root] (shape.rs:589-600) as an expression; see inside_subtree() ---- */
    std::string inside_fn(uint32_t first, uint32_t root) {
        char nm[64];
        snprintf(nm, sizeof nm, "in_%u_%u", first, root);
        if (inside_done.insert({first, root}).second) {
            Out d;
            d.f("    static EU_DEV bool %s(const EuScene &S, const real *p, uint32_t po = 0u, uint32_t bo = 0u) {\n", nm);
            std::vector<std::string> st;
            emit_inside_range(d, first, root, root, st);
            d.f("        return %s;\n    }\n", st.empty() ? "false" : st.back().c_str());
            inside_defs += d.s;
        }
        return nm;
    }
    /* walks ops[first..last] (a prefix of the subtree ending at `root`), leaves one bool variable name per finished subtree on `st` */
    void emit_inside_range(Out &d, uint32_t first, uint32_t last, uint32_t root, std::vector<std::string> &st) {
        for (uint32_t i = first; i <= last; i++) {
            const OpView p = op(i);
            char v[32];
            snprintf(v, sizeof v, "b%u", i);
            if (p.kind == EU_SH_SKIP) {
                /* guard of the bounded subtree ending at op p.first: a point outside its (enlarged) bounding sphere is in none of its
                 * leaves' solids by a margin that dwarfs rounding (trace_device.h inside_subtree: there the test is wave-uniform; per
                 * lane it gives the same answer by the same argument) */
                if (p.first > root) continue;      /* belongs to an enclosing subtree: not ours */
                std::vector<std::string> sub;
                snprintf(v, sizeof v, "bg%u", p.first);
                d.f("        bool %s = false;\n        { %s\n          if (!point_outside_bound<%d>(G, p)) {\n", v, bound_array("G", p.param).c_str(), D);
                emit_inside_range(d, i + 1, p.first, p.first, sub);
                d.f("          %s = %s;\n        } }\n", v, sub.empty() ? "false" : sub.back().c_str());
                st.push_back(v);
                i = p.first;
                continue;
            }
            if (p.kind >= EU_SH_CHAIN_UNION) {
                const char *call = p.kind == EU_SH_CHAIN_BOX ? "chain_inside_box<%d>(P, p)" : (p.kind == EU_SH_CHAIN_BOX0 && D == 3) ? "chain_inside_box<%d, true>(P, p)" : nullptr;
                d.f("        bool %s; { %s ", v, op_params("P", i).c_str());
                if (call) { d.f("%s = ", v); d.f(call, D); d.f("; }\n"); }
                else d.f("%s = chain_inside<%d>(%s, %uu, P, p); }\n", v, D, p.kind == EU_SH_CHAIN_UNION ? "true" : "false", p.count);
                st.push_back(v);
            } else if (p.kind < EU_SH_UNION) {
                d.f("        bool %s; { %s %s = leaf_inside<%d>(%s, P, p); }\n", v, op_params("P", i).c_str(), v, D, kind_name(p.kind));
                st.push_back(v);
            } else {
                if (st.size() < 2) { d.f("        /* malformed program at op %u */\n", i); continue; }
                const std::string b = st.back(); st.pop_back();
                const std::string a = st.back(); st.pop_back();
                const char *expr = p.kind == EU_SH_UNION ? "(%s | %s)" : p.kind == EU_SH_INTERSECTION ? "(%s & %s)" : p.kind == EU_SH_COMPLEMENT ? "(%s & !%s)" : "(%s ^ %s)";
                d.f("        const bool %s = ", v); d.f(expr, a.c_str(), b.c_str()); d.f(";\n");
                st.push_back(v);
            }
        }
    }

    /* A LEFT FOLD of congruent operands inside one shape program -- X1 X2 M X3 M ... Xk M, every X one leaf or chain of the same kind with its
     * parameters a constant stride after the one before, every M the same operation on (everything so far, X) -- is what ComposableShape::of
     * makes of a list (shape.rs:523-545): 4d_frame's four inner boxes, a union of spheres.  It gets ONE push and ONE merge in a loop over the
     * operands from the second on; the containment test of "everything so far" is the fold of the operands' tests in the same order.
     * Returns the number of operands (0: no such fold at op i), `stride` = reals between two operands' parameters. */
    bool no_folds = false;      /* tuning / tests: -DEU_JIT_NO_FOLDS */
    uint32_t fold_len(uint32_t i, uint32_t last, uint32_t root, uint32_t &stride) const {
        if (no_folds || !shape_params_in_memory || i + 2 > last) return 0;
        const OpView x1 = op(i), x2 = op(i + 1), m1 = op(i + 2);
        auto single = [](const OpView &x) { return x.kind != EU_SH_SKIP && x.kind != EU_SH_VOID && (x.kind < EU_SH_UNION || x.kind >= EU_SH_CHAIN_UNION); };
        auto merge = [](const OpView &m) { return m.kind >= EU_SH_UNION && m.kind < EU_SH_CHAIN_UNION; };
        if (!single(x1) || !single(x2) || !merge(m1) || x2.kind != x1.kind || x2.count != x1.count || x2.param <= x1.param) return 0;
        if (m1.first != i || m1.count != 0 || i + 2 >= root) return 0;
        stride = x2.param - x1.param;
        uint32_t k = 2, at = i + 3;      /* operands so far; the op after the last merge */
        while (at + 1 <= last && at + 1 < root) {
            const OpView x = op(at), m = op(at + 1);
            if (!single(x) || x.kind != x1.kind || x.count != x1.count || x.param != x1.param + k * stride) break;
            if (!merge(m) || m.kind != m1.kind || m.first != i || m.count != 0) break;
            k++; at += 2;
        }
        return k >= 3 ? k : 0;
    }
    std::string fold_fns;      /* static member functions of the folds: the operands' containment tests */
    void emit_fold(Out &d, uint32_t i, uint32_t k, uint32_t stride, std::vector<std::string> &st, const std::string &ind) {
        const OpView x1 = op(i), m1 = op(i + 2);
        const bool chain = x1.kind >= EU_SH_CHAIN_UNION;
        /* fin_<i>(S, p, j, po): operand j contains p;  fin_acc_<i>(S, p, n, po): the fold of operands 0..n-1 does (shape.rs:589-600, left to right) */
        {
            Out f;
            f.f("    static EU_DEV bool fin_%u(const EuScene &S, const real *p, uint32_t j, uint32_t po) {\n"
                "        const real *const P = (const real *)(S.w + %uu) + (%uu + j * %uu + po);\n", i, h->off_params, x1.param, stride);
            if (x1.kind == EU_SH_CHAIN_BOX) f.f("        return chain_inside_box<%d>(P, p);\n", D);
            else if (x1.kind == EU_SH_CHAIN_BOX0 && D == 3) f.f("        return chain_inside_box<%d, true>(P, p);\n", D);
            else if (chain) f.f("        return chain_inside<%d>(%s, %uu, P, p);\n", D, x1.kind == EU_SH_CHAIN_UNION ? "true" : "false", x1.count);
            else f.f("        return leaf_inside<%d>(%s, P, p);\n", D, kind_name(x1.kind));
            const char *acc = m1.kind == EU_SH_UNION ? "r = (r | b);" : m1.kind == EU_SH_INTERSECTION ? "r = (r & b);" : m1.kind == EU_SH_COMPLEMENT ? "r = (r & !b);" : "r = (r ^ b);";
            f.f("    }\n    static EU_DEV bool fin_acc_%u(const EuScene &S, const real *p, uint32_t n, uint32_t po) {\n"
                "        bool r = fin_%u(S, p, 0u, po);\n        _Pragma(\"nounroll\") for (uint32_t j = 1; j < n; j++) { const bool b = fin_%u(S, p, j, po); %s }\n        return r;\n    }\n", i, i, i, acc);
            fold_fns += f.s;
        }
        char acc[32];
        snprintf(acc, sizeof acc, "LF%u", i);
        const char *push = chain ? "push_chain" : "push_leaf";
        /* operand 0, then operands 1..k-1 in a loop: operand j >= 1 is op i + 2 j - 1 */
        if (chain) d.f("%sCsgList %s; { %s %s = push_chain<%d>(%s, %uu, P, o, d, hs, sp, %uu + oo, cnt, use_box, fail); }\n", ind.c_str(), acc, op_params("P", i).c_str(), acc, D, kind_name(x1.kind), x1.count, i);
        else d.f("%sCsgList %s; { %s %s = push_leaf<%d>(%s, P, o, d, hs, sp, %uu + oo, cnt); }\n", ind.c_str(), acc, op_params("P", i).c_str(), acc, D, kind_name(x1.kind), i);
        d.f("%s_Pragma(\"nounroll\") for (uint32_t fj = 1; fj < %uu; fj++) {   /* operands 1..%u of the fold that starts at op %u */\n", ind.c_str(), k, k - 1, i);
        d.f("%s    const real *const P = (const real *)(S.w + %uu) + (%uu + fj * %uu + po);\n", ind.c_str(), h->off_params, x1.param, stride);
        if (chain) d.f("%s    const CsgList Lb = %s<%d>(%s, %uu, P, o, d, hs, sp, %uu + 2u * fj - 1u + oo, cnt, use_box, fail);\n", ind.c_str(), push, D, kind_name(x1.kind), x1.count, i);
        else d.f("%s    const CsgList Lb = %s<%d>(%s, P, o, d, hs, sp, %uu + 2u * fj - 1u + oo, cnt);\n", ind.c_str(), push, D, kind_name(x1.kind), i);
        d.f("%s    %s = csg_merge<%d>(%s, false, hs, sp, %s, Lb, o, d, cnt, [&](const real *q) { return fin_acc_%u(S, q, fj, po); }, [&](const real *q) { return fin_%u(S, q, fj, po); });\n%s}\n",
            ind.c_str(), acc, D, kind_name(m1.kind), acc, i, i, ind.c_str());
        st.push_back(acc);
    }

    /* ---- eval_shape of ops[first..last] inside a tree, straight line; `st`: names of the CsgList variables on the hit stack ---- */
    void emit_tree_range(Out &d, uint32_t first, uint32_t last, uint32_t root, std::vector<std::string> &st, const std::string &ind) {
        for (uint32_t i = first; i <= last; i++) {
            const OpView p = op(i);
            char v[32];
            snprintf(v, sizeof v, "L%u", i);
            {
                uint32_t stride = 0;
                if (const uint32_t k = fold_len(i, last, root, stride)) {
                    emit_fold(d, i, k, stride, st, ind);
                    i += 2 * k - 2;      /* the fold's last merge is op i + 2 k - 2 */
                    continue;
                }
            }
            if (p.kind == EU_SH_SKIP) {      /* every ray of the wave misses the subtree's sphere -> its stream is empty (eval_shape) */
                snprintf(v, sizeof v, "LG%u", p.first);
                d.f("%sCsgList %s = {0u, false, false};\n%s{ %s\n%s  if (__ballot(!ray_misses_bound<%d>(G, o, d)) != 0ull) {\n", ind.c_str(), v, ind.c_str(),
                    bound_array("G", p.param).c_str(), ind.c_str(), D);
                std::vector<std::string> sub;
                emit_tree_range(d, i + 1, p.first, root, sub, ind + "    ");
                d.f("%s    %s = %s;\n%s} }\n", ind.c_str(), v, sub.empty() ? "CsgList{0u, false, false}" : sub.back().c_str(), ind.c_str());
                st.push_back(v);
                i = p.first;
                continue;
            }
            if (p.kind >= EU_SH_CHAIN_UNION) {
                d.f("%sCsgList %s; { %s %s = push_chain<%d>(%s, %uu, P, o, d, hs, sp, %uu + oo, cnt, use_box, fail); }\n", ind.c_str(), v, op_params("P", i).c_str(), v, D,
                    kind_name(p.kind), p.count, i);
                st.push_back(v);
            } else if (p.kind < EU_SH_UNION) {
                d.f("%sCsgList %s; { %s %s = push_leaf<%d>(%s, P, o, d, hs, sp, %uu + oo, cnt); }\n", ind.c_str(), v, op_params("P", i).c_str(), v, D, kind_name(p.kind), i);
                st.push_back(v);
            } else {
                if (st.size() < 2) { d.f("%s/* malformed program at op %u */\n", ind.c_str(), i); continue; }
                const OpView pb = op(i - 1);
                const uint32_t fb = pb.first, fa = p.first + p.count, ra = fb - 1, rb = i - 1;
                const std::string b = st.back(); st.pop_back();
                const std::string a = st.back(); st.pop_back();
                const std::string ia = inside_fn(fa, ra), ib = inside_fn(fb, rb);
                d.f("%sconst CsgList %s = csg_merge<%d>(%s, %s, hs, sp, %s, %s, o, d, cnt, [&](const real *q) { return %s(S, q, po, bo); }, [&](const real *q) { return %s(S, q, po, bo); });\n",
                    ind.c_str(), v, D, kind_name(p.kind), i == root ? "true" : "false", a.c_str(), b.c_str(), ia.c_str(), ib.c_str());
                st.push_back(v);
            }
        }
    }

    /* ---- LinearSpace expressions (material.rs:59-163): one RPN program as straight-line f64 statements; returns the result's name ---- */
    std::string emit_rpn(Out &d, uint64_t prog, const std::string &pre, const std::string &ind) {
        const uint32_t off = (uint32_t)prog, len = (uint32_t)(prog >> 32);
        std::vector<std::string> st;
        int tmp = 0;
        auto fresh = [&]() { char b[48]; snprintf(b, sizeof b, "%s_%d", pre.c_str(), tmp++); return std::string(b); };
        auto top = [&](size_t back) -> std::string { return st.size() > back ? st[st.size() - 1 - back] : std::string("0.0"); };
        for (uint32_t i = 0; i < len; i++) {
            const uint64_t wd = w[h->off_code + off + i];
            const uint32_t opc = (uint32_t)wd, arg = (uint32_t)(wd >> 32);
            switch (opc) {
            case EU_RPN_CONST: {
                i++;
                eu_f64 c; memcpy(&c, &w[h->off_code + off + i], 8);
                const std::string v = fresh();
                d.f("%sconst eu_f64 %s = %s;\n", ind.c_str(), v.c_str(), lit64(c).c_str());
                st.push_back(v);
                break;
            }
            case EU_RPN_VAR: {
                const std::string v = fresh();
                d.f("%sconst eu_f64 %s = ctx[%u];\n", ind.c_str(), v.c_str(), (arg >= 1 && (int)arg < D) ? arg : 0u);
                st.push_back(v);
                break;
            }
            case EU_RPN_NEG: {
                const std::string v = fresh();
                d.f("%sconst eu_f64 %s = -%s;\n", ind.c_str(), v.c_str(), top(0).c_str());
                if (!st.empty()) st.pop_back();
                st.push_back(v);
                break;
            }
            case EU_RPN_FN: {
                const bool binary = arg == EU_FN_MIN || arg == EU_FN_MAX || arg == EU_FN_ATAN2;
                const std::string y = top(0);
                if (!st.empty()) st.pop_back();
                std::string x = y;
                if (binary) { x = top(0); if (!st.empty()) st.pop_back(); }
                const std::string v = fresh();
                const char *fn1 = nullptr, *fn2 = nullptr;
                switch (arg) {
                case EU_FN_SQRT: fn1 = "sqrt"; break; case EU_FN_ABS: fn1 = "fabs"; break; case EU_FN_FLOOR: fn1 = "floor"; break;
                case EU_FN_CEIL: fn1 = "ceil"; break; case EU_FN_MIN: fn2 = "rpn_min"; break; case EU_FN_MAX: fn2 = "rpn_max"; break;
                case EU_FN_SIN: fn1 = "eu_sin_f64"; break; case EU_FN_COS: fn1 = "eu_cos_f64"; break; case EU_FN_TAN: fn1 = "eu_tan_f64"; break;
                case EU_FN_ASIN: fn1 = "eu_asin_f64"; break; case EU_FN_ACOS: fn1 = "eu_acos_f64"; break; case EU_FN_ATAN: fn1 = "eu_atan_f64"; break;
                case EU_FN_ATAN2: fn2 = "eu_atan2_f64"; break; default: fn1 = "rpn_signum"; break;
                }
                if (fn2) d.f("%sconst eu_f64 %s = %s(%s, %s);\n", ind.c_str(), v.c_str(), fn2, x.c_str(), y.c_str());
                else d.f("%sconst eu_f64 %s = %s((eu_f64)%s);\n", ind.c_str(), v.c_str(), fn1, x.c_str());
                st.push_back(v);
                break;
            }
            default: {
                const std::string y = top(0);
                if (!st.empty()) st.pop_back();
                const std::string x = top(0);
                if (!st.empty()) st.pop_back();
                const std::string v = fresh();
                switch (opc) {
                case EU_RPN_ADD: d.f("%sconst eu_f64 %s = %s + %s;\n", ind.c_str(), v.c_str(), x.c_str(), y.c_str()); break;
                case EU_RPN_SUB: d.f("%sconst eu_f64 %s = %s - %s;\n", ind.c_str(), v.c_str(), x.c_str(), y.c_str()); break;
                case EU_RPN_MUL: d.f("%sconst eu_f64 %s = %s * %s;\n", ind.c_str(), v.c_str(), x.c_str(), y.c_str()); break;
                case EU_RPN_DIV: d.f("%sconst eu_f64 %s = %s / %s;\n", ind.c_str(), v.c_str(), x.c_str(), y.c_str()); break;
                case EU_RPN_REM: d.f("%sconst eu_f64 %s = fmod((eu_f64)%s, (eu_f64)%s);\n", ind.c_str(), v.c_str(), x.c_str(), y.c_str()); break;
                default: d.f("%sconst eu_f64 %s = pow_int(%s, %s);\n", ind.c_str(), v.c_str(), x.c_str(), y.c_str()); break;
                }
                st.push_back(v);
                break;
            }
            }
        }
        return st.empty() ? std::string("0.0") : st.front();      /* eval_rpn returns the BOTTOM of its stack */
    }

    /* Material::enter (exit_ = false) / exit of material m (material.rs:135-162) on `dir` */
    void emit_material(Out &d, uint32_t m, bool exit_, const std::string &ind) {
        const uint64_t mw = w[h->off_materials + m];
        const uint32_t kind = (uint32_t)mw & 0xff, ntr = ((uint32_t)mw >> 8), first = (uint32_t)(mw >> 32);
        if (kind != EU_MAT_LINEAR) return;
        for (uint32_t k = 0; k < ntr; k++) {
            const uint32_t tr = exit_ ? (first + ntr - 1 - k) : (first + k);
            d.f("%s{ eu_f64 ctx[%d];\n", ind.c_str(), D);      /* the evaluation context is the vector BEFORE the transformation (material.rs:99-111) */
            for (int i = 0; i < D; i++) d.f("%s  ctx[%d] = dir[%d];\n", ind.c_str(), i, i);
            for (int i = 0; i < D; i++) {
                char pre[32];
                snprintf(pre, sizeof pre, "t%u_%d", tr, i);
                const std::string r = emit_rpn(d, w[h->off_transforms + 8 * tr + (exit_ ? 4 : 0) + (uint32_t)i], pre, ind + "  ");
                d.f("%s  dir[%d] = (real)%s;\n", ind.c_str(), i, r.c_str());
            }
            d.f("%s}\n", ind.c_str());
        }
    }

    std::string mapped_record(const std::string &name, uint32_t id) const {
        const EuFlatMapped *M = mapped(id);
        Out d;
        d.f("static constexpr EuFlatMapped %s = {%uu, %uu, %uu, %uu, 0ull, {%s, %s, %s}, %s, %s};", name.c_str(), M->tex_kind, M->uv_kind, M->w, M->h,
            lit64(M->center[0]).c_str(), lit64(M->center[1]).c_str(), lit64(M->center[2]).c_str(), lit64(M->wd).c_str(), lit64(M->hd).c_str());
        return d.s;
    }

    void generate() {
        const uint32_t ne = h->n_entities;
        o.f("/* generated by euclider_amd (%s): trace kernels specialised for one scene: %u entities, %u shape ops, dim %d (budgets: %u ops, %u surfaces) */\n", EU_JIT_VERSION, ne, h->n_ops, D,
            ops_budget, surfaces_budget);
        /* tuning of the specialised shade kernel (measured with `--jit-flags`, one call per sweep): windows of 2048 rays sorted together
         * where there are surfaces to sort by (three or more; the kernel halves and quarters its windows by itself when a launch is
         * small; 4d_frame, one surface, loses 1 % with the larger window) and launch bounds of two waves per SIMD -- the
         * kernel still runs three (129 VGPRs), the compiler is only freed from the 168-register limit: 3d_room 7.4 -> 7.7 Gray/s,
         * 3d_hallways / 4d_frame / 4d_cylinders within +-1 %.  The ahead-of-time interpreter kernels keep 1024 / 3. */
        o.f("#ifndef EU_WF_WIN\n#define EU_WF_WIN %u\n#endif\n#ifndef EU_SHADE_WAVES\n#define EU_SHADE_WAVES 2\n#endif\n", h->n_surfaces >= 3 ? 2048u : 1024u);
        o.f("#include \"trace_wavefront.h\"\n\n");
        o.f("template <int D> EU_DEV bool point_outside_bound(const real *Bd, const real *p) {\n    real rr = R(0.0);\n#pragma unroll\n"
            "    for (int m = 0; m < D; m++) { const real q = p[m] - Bd[m]; rr = rr + q * q; }\n    return rr > Bd[D];\n}\n\n");

        /* ---- trace_closest ---- */
        Out tc;
        tc.f("    /* trace_closest (universe/mod.rs:85-147): every surfaced entity's shape program as a straight line */\n");
        tc.f("    template <class HS>\n    static EU_DEV void trace_closest(const EuScene &S, const real *o, const real *d, HS &hs, LaneCounters &cnt, int use_box, bool &fail,\n"
             "                                     bool &have, real &best_t, uint32_t &best_code, uint32_t &best_ent) {\n");
        straight.assign(ne, 0);
        find_runs();
        for (uint32_t e = 0; e < ne; e++) {
            const EntityView E = entity(e);
            const uint32_t n = E.shape_root - E.shape_first + 1u;
            const uint32_t members = run_of[e] >= 0 ? runs[(size_t)run_of[e]].count : 1u;      /* (a run's body is written once) */
            if (n_straight_ops + n <= ops_budget) { for (uint32_t k = 0; k < members; k++) straight[e + k] = 1; n_straight_ops += n; }
            else n_interp_entities += members;
            e += members - 1u;
        }
        for (uint32_t e = 0; e < ne; e++) {
            const EntityView E = entity(e);
            if (E.surface < 0) continue;
            const Run *run = (straight[e] && run_of[e] >= 0) ? &runs[(size_t)run_of[e]] : nullptr;
            if (!straight[e]) {      /* this entity and what follows it, up to the next one with code of its own */
                uint32_t e2 = e + 1;
                while (e2 < ne && (entity(e2).surface < 0 || !straight[e2])) e2++;
                tc.f("        interp_entities<%d>(S, %uu, %uu, o, d, hs, cnt, use_box, fail, have, best_t, best_code, best_ent);\n", D, e, e2);
                e = e2 - 1;
                continue;
            }
            if (run) {
                tc.f("        _Pragma(\"nounroll\") for (uint32_t ge = 0; ge < %uu; ge++) {   /* entities %u..%u: congruent, ops %u..%u each */\n"
                     "            const uint32_t po = ge * %uu, bo = ge * %uu, oo = ge * %uu;\n", run->count, e, e + run->count - 1u, E.shape_first, E.shape_root,
                     run->dp, run->db * (uint32_t)(D + 2), run->n_ops);
            } else {
                tc.f("        {   /* entity %u: ops %u..%u */\n            constexpr uint32_t po = 0u, bo = 0u, oo = 0u;\n", e, E.shape_first, E.shape_root);
            }
            std::string ind = "            ";
            if (E.bound != 0xffffffffu) { tc.f("            %s\n            if (!ray_misses_bound<%d>(B, o, d)) {\n", bound_array("B", E.bound).c_str(), D); ind += "    "; }
            tc.f("%sreal t = R(0.0); uint32_t code = 0, n;\n", ind.c_str());
            if (E.shape_first == E.shape_root) {
                const OpView p = op(E.shape_root);
                tc.f("%s{ %s n = eval_single<%d>(%s, %uu, P, o, d, hs, %uu + oo, cnt, t, code, use_box, fail); }\n", ind.c_str(), op_params("P", E.shape_root).c_str(), D,
                     kind_name(p.kind), p.count, E.shape_root);
            } else {
                tc.f("%s{ uint32_t sp = 0;\n", ind.c_str());
                std::vector<std::string> st;
                emit_tree_range(tc, E.shape_first, E.shape_root, E.shape_root, st, ind + "  ");
                tc.f("%s  n = csg_root_result(%s, hs, cnt, t, code); }\n", ind.c_str(), st.empty() ? "CsgList{0u, false, false}" : st.back().c_str());
            }
            tc.f("%sif (n != 0 && (!have || best_t > t)) { have = true; best_t = t; best_code = code; best_ent = %uu%s; }\n", ind.c_str(), e, run ? " + ge" : "");
            if (E.bound != 0xffffffffu) tc.f("            }\n");
            tc.f("        }\n");
            if (run) e += run->count - 1u;
        }
        tc.f("    }\n");

        /* ---- hit_normal ---- */
        Out hn;
        hn.f("    static EU_DEV void hit_normal(const EuScene &S, uint32_t ent, uint32_t code, const real *o, const real *d, const real *loc, real *n) {\n");
        {
            bool any_run = false;
            for (const Run &r : runs) any_run = any_run || straight[r.e0];
            if (!any_run) hn.f("        const uint32_t opi = code & 0xffffu;\n        constexpr uint32_t po = 0u;\n");
            else {      /* a leaf of a run's entity k: the case of the run's first entity, its parameters k strides further on */
                hn.f("        uint32_t opi = code & 0xffffu, po = 0u;\n");
                for (const Run &r : runs) {
                    if (!straight[r.e0]) continue;
                    const uint32_t f0 = entity(r.e0).shape_first;
                    hn.f("        if (opi - %uu < %uu) { const uint32_t ge = (opi - %uu) / %uu; opi -= ge * %uu; po = ge * %uu; }\n", f0, r.count * r.n_ops, f0, r.n_ops, r.n_ops, r.dp);
                }
            }
        }
        hn.f("        switch (opi) {\n");
        for (uint32_t e = 0; e < ne; e++) {
            const EntityView E = entity(e);
            if (E.surface < 0 || !straight[e]) continue;
            const bool in_run = run_of[e] >= 0;
            if (in_run && runs[(size_t)run_of[e]].e0 != e) continue;      /* (written once, for the run's first entity) */
            for (uint32_t i = E.shape_first; i <= E.shape_root; i++) {
                const OpView p = op(i);
                if (p.kind == EU_SH_SKIP || (p.kind >= EU_SH_UNION && p.kind < EU_SH_CHAIN_UNION) || p.kind == EU_SH_VOID) continue;
                if (p.kind >= EU_SH_CHAIN_UNION && in_run) {      /* (no table of constants: the normals differ from entity to entity) */
                    hn.f("        case %uu: { %s const real *const Q = P + ((code >> 16) & 0xffu) * %uu;\n", i, op_params("P", i).c_str(), 2u * (uint32_t)D + 2u);
                    for (int m = 0; m < D; m++) hn.f("            n[%d] = Q[%d];\n", m, D + 2 + m);
                    hn.f("            break; }\n");
                } else
                if (p.kind >= EU_SH_CHAIN_UNION) {      /* a chain's leaf counts as a half-space: its normal is the stored n * -signum (shape.rs:860) */
                    const real *P = params(p.param);
                    hn.f("        case %uu: { static constexpr real NF[%u][%d] = {", i, p.count, D);
                    for (uint32_t k = 0; k < p.count; k++) {
                        hn.f("%s{", k ? ", " : "");
                        for (int m = 0; m < D; m++) hn.f("%s%s", m ? ", " : "", litr(P[k * (2 * (uint32_t)D + 2) + (uint32_t)D + 2 + (uint32_t)m]).c_str());
                        hn.f("}");
                    }
                    hn.f("};\n            const uint32_t k = (code >> 16) & 0xffu;\n");
                    for (int m = 0; m < D; m++) hn.f("            n[%d] = NF[k][%d];\n", m, m);
                    hn.f("            break; }\n");
                } else {
                    hn.f("        case %uu: { %s leaf_normal<%d>(%s, P, o, d, loc, n); break; }\n", i, op_params("P", i).c_str(), D, kind_name(p.kind));
                }
            }
        }
        if (n_interp_entities) hn.f("        default: ::hit_normal<%d>(S, code, o, d, loc, n); return;      /* a leaf of an entity without code of its own */\n", D);
        else hn.f("        default: break;\n");
        hn.f("        }\n        if (code & EU_HIT_FLIP) {\n");
        for (int m = 0; m < D; m++) hn.f("            n[%d] = -n[%d];\n", m, m);
        hn.f("        }\n    }\n");

        /* ---- surfaces ---- */
        Out sf;
        std::map<int32_t, std::vector<uint32_t>> by_surface;
        for (uint32_t e = 0; e < ne; e++) { const EntityView E = entity(e); if (E.surface >= 0) by_surface[E.surface].push_back(e); }
        std::set<int32_t> own_surface;      /* surfaces with a function of their own: in the order their first entity comes */
        for (uint32_t e = 0; e < ne; e++) {
            const EntityView E = entity(e);
            if (E.surface >= 0 && !own_surface.count(E.surface) && own_surface.size() < surfaces_budget) own_surface.insert(E.surface);
        }
        n_generic_surfaces = (uint32_t)(by_surface.size() - own_surface.size());
        for (auto &kv : by_surface) {
            if (!own_surface.count(kv.first)) continue;
            const EuFlatSurface *F = surface((uint32_t)kv.first);
            sf.f("    static EU_DEV void surf_%d(const EuScene &S, HitCtx<%d> &c, real time_s, LaneCounters &cnt, SurfaceEval<%d> &E) {\n", kv.first, D, D);
            sf.f("        static constexpr EuFlatSurface F = {%uu, %uu, %uu, %uu, %s, %s, %s, %s, {0.0, 0.0}};\n", F->ratio_kind, F->thr_kind, F->color_first, F->color_root,
                 lit64(F->ratio_p0).c_str(), lit64(F->ratio_p1).c_str(), lit64(F->thr_p0).c_str(), lit64(F->thr_p0_inv).c_str());
            sf.f("        surface_eval<%d>(&F, c, cnt, E, [&]() -> Rgba {\n", D);
            std::vector<std::string> st;
            for (uint32_t i = F->color_first; i <= F->color_root; i++) {
                const EuFlatColorOp *C = color_op(i);
                char v[32], cn[32];
                snprintf(v, sizeof v, "v%u", i);
                snprintf(cn, sizeof cn, "C%u", i);
                sf.f("            static constexpr EuFlatColorOp %s = {%uu, %uu, %uu, 0u, {%s, %s, %s, %s}, {%s, %s, %s, %s}, {%s, %s, %s, %s}, {0.0, 0.0}};\n", cn, C->kind, C->fn, C->aux,
                     lit64(C->c0[0]).c_str(), lit64(C->c0[1]).c_str(), lit64(C->c0[2]).c_str(), lit64(C->c0[3]).c_str(),
                     lit64(C->c1[0]).c_str(), lit64(C->c1[1]).c_str(), lit64(C->c1[2]).c_str(), lit64(C->c1[3]).c_str(),
                     lit64(C->v[0]).c_str(), lit64(C->v[1]).c_str(), lit64(C->v[2]).c_str(), lit64(C->v[3]).c_str());
                switch (C->kind) {
                case EU_COL_UNIFORM: sf.f("            const Rgba %s = col_uniform(&%s);\n", v, cn); break;
                case EU_COL_BLEND: {
                    std::string dst = st.empty() ? "Rgba{}" : st.back(); if (!st.empty()) st.pop_back();
                    std::string src = st.empty() ? "Rgba{}" : st.back(); if (!st.empty()) st.pop_back();
                    sf.f("            const Rgba %s = col_blend(&%s, %s, %s);\n", v, cn, src.c_str(), dst.c_str());
                    break;
                }
                case EU_COL_ILLUM_GLOBAL: sf.f("            const Rgba %s = col_illum_global<%d>(&%s, c);\n", v, D, cn); break;
                case EU_COL_ILLUM_DIR: sf.f("            const Rgba %s = col_illum_dir<%d>(&%s, c);\n", v, D, cn); break;
                case EU_COL_PERLIN: sf.f("            const Rgba %s = col_perlin<%d>(&%s, S.perlin(%uu), c, time_s);\n", v, D, cn, C->aux); break;
                default: sf.f("            %s\n            const Rgba %s = mapped_get_color(&M%u, S.texels(%uu), c.loc, cnt);\n", mapped_record("M" + std::to_string(i), C->aux).c_str(), v, i, C->aux); break;
                }
                st.push_back(v);
            }
            sf.f("            return %s;\n        });\n    }\n", st.empty() ? "Rgba{}" : st.back().c_str());
        }
        sf.f("    static EU_DEV void surface(const EuScene &S, uint32_t ent, HitCtx<%d> &c, real time_s, LaneCounters &cnt, real *cst, uint32_t stride, SurfaceEval<%d> &E) {\n"
             "        switch (ent) {\n", D, D);
        for (auto &kv : by_surface) {
            if (!own_surface.count(kv.first)) continue;
            sf.f("       ");
            for (uint32_t e : kv.second) sf.f(" case %uu:", e);
            sf.f(" surf_%d(S, c, time_s, cnt, E); break;\n", kv.first);
        }
        if (n_generic_surfaces)
            sf.f("        default: {      /* a surface without a function of its own: from its records, as the interpreter kernels do */\n"
                 "            const EuFlatSurface *F = S.surface((uint32_t)S.entity(ent).surface);\n"
                 "            surface_eval<%d>(F, c, cnt, E, [&]() { return surface_color<%d>(S, F, c, time_s, cnt, cst, stride); });\n            break; }\n", D, D);
        else
        sf.f("        default: E.ratio = R(0.0); E.have_color = false; E.translucent = false; E.spx = 0; E.sc = Rgba{R(0.0), R(0.0), R(0.0), R(0.0)}; break;\n");
        sf.f("        }\n    }\n");

        /* ---- material_at (universe/mod.rs:229-251): first entity containing the point ---- */
        Out ma;
        ma.f("    static EU_DEV int material_at(const EuScene &S, const real *p) {\n");
        for (uint32_t e = 0; e < ne; e++) {
            const EntityView E = entity(e);
            if (straight[e] && run_of[e] >= 0) {
                const Run &r = runs[(size_t)run_of[e]];
                ma.f("        _Pragma(\"nounroll\") for (uint32_t ge = 0; ge < %uu; ge++) if (%s(S, p, ge * %uu, ge * %uu)) return (int)(%uu + ge);\n", r.count,
                     inside_fn(E.shape_first, E.shape_root).c_str(), r.dp, r.db * (uint32_t)(D + 2), e);
                e += r.count - 1u;
                continue;
            }
            if (straight[e]) { ma.f("        if (%s(S, p)) return %u;\n", inside_fn(E.shape_first, E.shape_root).c_str(), e); continue; }
            uint32_t e2 = e + 1;
            while (e2 < ne && !straight[e2]) e2++;
            ma.f("        { const int r = material_at_range<%d>(S, %uu, %uu, p); if (r >= 0) return r; }\n", D, e, e2);
            e = e2 - 1;
        }
        ma.f("        return -1;\n    }\n");

        /* ---- Material::enter / exit of the material of entity `ent` ---- */
        Out mp;
        std::map<uint32_t, std::vector<uint32_t>> by_material;
        for (uint32_t e = 0; e < ne; e++) by_material[entity(e).material].push_back(e);
        mp.f("    static EU_DEV void material_apply(const EuScene &S, uint32_t ent, real *dir, bool exit_) {\n        switch (ent) {\n");
        for (auto &kv : by_material) {
            const uint64_t mw = w[h->off_materials + kv.first];
            if (((uint32_t)mw & 0xff) != EU_MAT_LINEAR) continue;      /* Vacuum: enter and exit leave the direction alone (material.rs:32-57) */
            mp.f("       ");
            for (uint32_t e : kv.second) mp.f(" case %uu:", e);
            mp.f("\n            if (!exit_) {\n");
            emit_material(mp, kv.first, false, "                ");
            mp.f("            } else {\n");
            emit_material(mp, kv.first, true, "                ");
            mp.f("            }\n            break;\n");
        }
        mp.f("        default: break;\n        }\n    }\n");

        /* ---- background ---- */
        Out bg;
        bg.f("    static EU_DEV Rgba background(const EuScene &S, const real *point, LaneCounters &cnt) {\n        %s\n        return mapped_get_color(&M, S.texels(%uu), point, cnt);\n    }\n",
             mapped_record("M", h->background).c_str(), h->background);

        o.f("struct EuJit {\n    static constexpr bool kInterpreter = false;\n");
        o.s += inside_defs;      /* (complete by now: every emitter above has registered the subtrees it tests) */
        o.s += fold_fns;
        o.s += tc.s; o.s += hn.s; o.s += sf.s; o.s += ma.s; o.s += mp.s; o.s += bg.s;
        o.f("};\n\n");
    }
};

static std::string hex_digest(const std::string &a, const std::string &b) {
    auto fnv = [](uint64_t seed, const std::string &s, uint64_t hsh) { hsh ^= seed; for (unsigned char c : s) { hsh ^= c; hsh *= 0x100000001b3ull; } return hsh; };
    uint64_t h1 = fnv(0x1234567ull, a, 0xcbf29ce484222325ull), h2 = fnv(0x9e3779b97f4a7c15ull, a, 0x84222325cbf29ce4ull);
    h1 = fnv(1, b, h1); h2 = fnv(2, b, h2);
    char buf[40];
    snprintf(buf, sizeof buf, "%016llx%016llx", (unsigned long long)h1, (unsigned long long)h2);
    return buf;
}

}  // namespace

JitPlan jit_generate(const FlatScene &flat, const std::string &extra_flags, bool fused) {
    JitPlan plan;
    plan.fused = fused;
    {
        size_t i = 0;
        while (i < extra_flags.size()) {
            while (i < extra_flags.size() && extra_flags[i] == ' ') i++;
            size_t j = i;
            while (j < extra_flags.size() && extra_flags[j] != ' ') j++;
            if (j > i) plan.extra_flags.push_back(extra_flags.substr(i, j - i));
            i = j;
        }
    }
    const EuFlatHeader &h = flat.header();
    plan.dim = (int)h.dim;
    /* the per-lane hit stack: in LDS while three workgroups per CU still fit (4 waves x cap x 64 lanes x (sizeof(real) + 4) bytes each),
     * else a private array of exactly the entries this scene needs (the ahead-of-time kernels only have 16 and 96) */
    const uint32_t soft_cap = h.hit_cap & 0xffffu;      /* (bits 16..: the strict bound of the stack kernels, flat_scene.h) */
    const uint32_t cap = soft_cap < 8 ? 8u : ((soft_cap + 3u) & ~3u);
    plan.hs_lds = cap <= 24;      /* two workgroups per CU at least: 24 entries x 64 lanes x (sizeof(real) + 4) B x 4 waves = 72 KB */
    plan.hs_cap = cap;
    for (auto &f : plan.extra_flags)      /* tuning experiments (bench.py --jit-flags=-DEU_HS_CAP=16): a smaller stack than the static bound; a lane that needs more marks the frame (EU_CNT_HS_FULL) */
        if (f == "-DEU_HS_PRIVATE") plan.hs_lds = false;
        else if (f.rfind("-DEU_HS_CAP=", 0) == 0) { const unsigned v = (unsigned)atoi(f.c_str() + 12); if (v >= 4 && v <= 96) { plan.hs_cap = v; plan.hs_lds = v <= 24; } }
    Gen g(flat);
    for (auto &f : plan.extra_flags)      /* (tests and tuning: small budgets put every scene on the mixed path) */
        if (f.rfind("-DEU_JIT_OPS_BUDGET=", 0) == 0) g.ops_budget = (uint32_t)strtoul(f.c_str() + 20, nullptr, 10);
        else if (f.rfind("-DEU_JIT_SURFACES_BUDGET=", 0) == 0) g.surfaces_budget = (uint32_t)strtoul(f.c_str() + 25, nullptr, 10);
        else if (f == "-DEU_JIT_NO_RUNS") g.no_runs = true;
        else if (f == "-DEU_JIT_NO_FOLDS") g.no_folds = true;
    g.generate();
    plan.n_straight_ops = g.n_straight_ops;
    plan.n_interp_entities = g.n_interp_entities;
    plan.n_generic_surfaces = g.n_generic_surfaces;
    plan.color_stack = g.n_generic_surfaces != 0;
    Out tail;
    const unsigned hscap = plan.hs_lds ? 0u : plan.hs_cap;
    tail.f("extern \"C\" __global__ __launch_bounds__(EU_WF_BLOCK, EU_ISECT_WAVES) void eu_jit_intersect0(const uint64_t *__restrict__ scene_g, uint32_t hs_cap, EuDevCamera cam, EuDevFrame fr,\n"
           "        EuWfBuffers B, EuDevCounters *counters, eu_f64 *__restrict__ hit_t_aov) {\n    extern __shared__ uint64_t lds_dyn[];\n"
           "    wf_intersect_body<%d, %u, EuJit, true>(scene_g, hs_cap, 0u, cam, fr, B, counters, hit_t_aov, lds_dyn);\n}\n\n", plan.dim, hscap);
    if (!fused) {
    tail.f("extern \"C\" __global__ __launch_bounds__(EU_WF_BLOCK, EU_ISECT_WAVES) void eu_jit_intersect(const uint64_t *__restrict__ scene_g, uint32_t hs_cap, uint32_t gen,\n"
           "        EuWfBuffers B, EuDevCounters *counters) {\n    extern __shared__ uint64_t lds_dyn[];\n    const EuDevCamera cam = {};\n    const EuDevFrame fr = {};\n"
           "    wf_intersect_body<%d, %u, EuJit, false>(scene_g, hs_cap, gen, cam, fr, B, counters, nullptr, lds_dyn);\n}\n\n", plan.dim, hscap);
    tail.f("extern \"C\" __global__ __launch_bounds__(EU_WF_BLOCK, EU_SHADE_WAVES) void eu_jit_shade(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t gen, uint32_t max_depth, real time_s,\n"
           "        EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ point_rgb) {\n    extern __shared__ uint64_t lds_dyn[];\n    const EuDevCamera cam = {};\n    const EuDevFrame fr = {};\n"
           "    wf_shade_body<%d, false, EuJit, false>(scene_g, scene_words, gen, max_depth, time_s, cam, fr, B, counters, rgba, nullptr, point_rgb, lds_dyn);\n}\n\n", plan.dim);
    tail.f("extern \"C\" __global__ __launch_bounds__(EU_WF_BLOCK, EU_SHADE_WAVES) void eu_jit_shade0(const uint64_t *__restrict__ scene_g, uint32_t scene_words, EuDevCamera cam, EuDevFrame fr,\n"
           "        EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ hit_t_aov, eu_f64 *__restrict__ point_rgb) {\n    extern __shared__ uint64_t lds_dyn[];\n"
           "    wf_shade_body<%d, false, EuJit, true>(scene_g, scene_words, 0u, cam.max_depth, fr.time_s, cam, fr, B, counters, rgba, hit_t_aov, point_rgb, lds_dyn);\n}\n", plan.dim);
    } else {
    /* the fused forms: shade generation g, then intersect the rays just queued (trace_wavefront.h: FUSE) */
    tail.f("#ifndef EU_FSHADE_WAVES\n#define EU_FSHADE_WAVES EU_SHADE_WAVES\n#endif\n");
    tail.f("extern \"C\" __global__ __launch_bounds__(EU_WF_BLOCK, EU_FSHADE_WAVES) void eu_jit_fshade(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t hs_cap, uint32_t gen, uint32_t max_depth, real time_s,\n"
           "        EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ point_rgb) {\n    extern __shared__ uint64_t lds_dyn[];\n    const EuDevCamera cam = {};\n    const EuDevFrame fr = {};\n"
           "    wf_shade_body<%d, false, EuJit, false, %u>(scene_g, scene_words, gen, max_depth, time_s, cam, fr, B, counters, rgba, nullptr, point_rgb, lds_dyn, hs_cap);\n}\n\n", plan.dim, hscap);
    tail.f("extern \"C\" __global__ __launch_bounds__(EU_WF_BLOCK, EU_SHADE_WAVES) void eu_jit_fshade0(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t hs_cap, EuDevCamera cam, EuDevFrame fr,\n"
           "        EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ hit_t_aov, eu_f64 *__restrict__ point_rgb) {\n    extern __shared__ uint64_t lds_dyn[];\n"
           "    wf_shade_body<%d, false, EuJit, true, %u>(scene_g, scene_words, 0u, cam.max_depth, fr.time_s, cam, fr, B, counters, rgba, hit_t_aov, point_rgb, lds_dyn, hs_cap);\n}\n", plan.dim, hscap);
    }
    plan.source = g.o.s + tail.s;
    std::string dep = EU_JIT_VERSION;
    {   /* a code object is only as good as the compiler that made it: the hiprtc version is part of the key */
        int major = 0, minor = 0;
        if (hiprtcVersion(&major, &minor) == HIPRTC_SUCCESS) dep += " hiprtc " + std::to_string(major) + "." + std::to_string(minor);
    }
    for (const char *f : kCompileFlags) { dep += ' '; dep += f; }
    for (const std::string &f : plan.extra_flags) { dep += ' '; dep += f; }
    for (int k = 0; k < kNumHeaders; k++) { dep += kHeaders[k].name; dep += kHeaders[k].text; }
    plan.key = hex_digest(plan.source, dep);
    return plan;
}

/* ------------------------------------------------------------------ build + cache */
static std::mutex g_mem_mutex;
static std::map<std::string, std::vector<char>> g_mem_cache;      /* key -> code object (a process often creates several renderers of one scene) */
static std::deque<std::string> g_mem_order;                       /* oldest first: a long-lived process that loads scene after scene keeps the last 64 */
static void mem_cache_put(const std::string &fname, const std::vector<char> &code) {      /* (g_mem_mutex held) */
    if (g_mem_cache.find(fname) == g_mem_cache.end()) g_mem_order.push_back(fname);
    g_mem_cache[fname] = code;
    while (g_mem_order.size() > 64) { g_mem_cache.erase(g_mem_order.front()); g_mem_order.pop_front(); }
}

static bool read_file(const std::string &path, std::vector<char> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<char> buf;
    char tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    if (buf.size() < 64 || memcmp(buf.data(), "\x7f" "ELF", 4) != 0) return false;      /* a code object is an ELF file */
    out.swap(buf);
    return true;
}

static void mkdirs(const std::string &dir) {
    std::string cur;
    for (size_t i = 0; i <= dir.size(); i++) {
        if (i == dir.size() || dir[i] == '/') { if (!cur.empty()) (void)mkdir(cur.c_str(), i == dir.size() ? 0700 : 0755); }
        if (i < dir.size()) cur += dir[i];
    }
}

/* Code objects run on the GPU as they are found: a cache directory is only read or written if it belongs to this user and nobody else
 * can write to it (a world-writable directory would let another local user plant <key>.hsaco). */
static bool dir_is_private(const std::string &dir) {
    struct stat st;
    if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}

static std::string default_cache_dir() {      /* "": no private place to keep code objects -- they then live in memory only */
    if (const char *x = getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/euclider_amd";
    if (const char *hm = getenv("HOME")) if (*hm) return std::string(hm) + "/.cache/euclider_amd";
    return std::string();
}

static std::string library_cache_dir() {      /* <directory of this shared library>/jit_cache */
    Dl_info info;
    if (dladdr((const void *)&default_cache_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t s = p.rfind('/');
        if (s != std::string::npos) return p.substr(0, s) + "/jit_cache";
    }
    return std::string();
}

/* one hiprtc compilation of the plan's source (extra: one more flag, or nullptr) */
static int hiprtc_compile(const JitPlan &plan, const char *extra, std::vector<char> &code, std::string &log) {
    hiprtcProgram prog = nullptr;
    std::vector<const char *> hdr_text, hdr_name;
    for (int k = 0; k < kNumHeaders; k++) { hdr_text.push_back(kHeaders[k].text); hdr_name.push_back(kHeaders[k].name); }
    hiprtcResult rc = hiprtcCreateProgram(&prog, plan.source.c_str(), "eu_jit_scene.hip", kNumHeaders, hdr_text.data(), hdr_name.data());
    if (rc != HIPRTC_SUCCESS) { log = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(rc); return EU_ERR_HIP; }
    std::vector<const char *> flags(kCompileFlags, kCompileFlags + sizeof(kCompileFlags) / sizeof(kCompileFlags[0]));
    for (const std::string &f : plan.extra_flags) flags.push_back(f.c_str());
    if (extra) flags.push_back(extra);
    rc = hiprtcCompileProgram(prog, (int)flags.size(), flags.data());
    size_t log_size = 0;
    if (hiprtcGetProgramLogSize(prog, &log_size) == HIPRTC_SUCCESS && log_size > 1) {
        log.resize(log_size);
        (void)hiprtcGetProgramLog(prog, &log[0]);
    }
    if (rc != HIPRTC_SUCCESS) {
        log = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(rc) + "\n" + log;
        (void)hiprtcDestroyProgram(&prog);
        return EU_ERR_HIP;
    }
    size_t code_size = 0;
    rc = hiprtcGetCodeSize(prog, &code_size);
    if (rc == HIPRTC_SUCCESS) { code.resize(code_size); rc = hiprtcGetCode(prog, code.data()); }
    (void)hiprtcDestroyProgram(&prog);
    if (rc != HIPRTC_SUCCESS || code_size == 0) { log = std::string("hiprtcGetCode: ") + hiprtcGetErrorString(rc); return EU_ERR_HIP; }
    return EU_OK;
}

/* A count out of a code object's metadata (the NT_AMDGPU_METADATA note: MessagePack, every kernel a map with its keys in alphabetical
 * order, ".name" before ".vgpr_count" and ".vgpr_spill_count"): the value of `key` in the map of kernel `kernel`; -1 if not found.
 * (Not a MessagePack reader: the byte patterns of "key: string" and "key: small unsigned" are looked for, which is all it is used for.) */
static int code_object_count(const std::vector<char> &code, const char *kernel, const char *key) {
    auto str = [](const char *t) { std::string o; const size_t n = strlen(t); o.push_back((char)(n < 32 ? 0xa0 + n : 0xd9)); if (n >= 32) o.push_back((char)n); o += t; return o; };
    const std::string name = str(".name") + str(kernel), want = str(key), next_kernel = str(".name");
    const std::string blob(code.begin(), code.end());
    const size_t at = blob.find(name);
    if (at == std::string::npos) return -1;
    const size_t end = blob.find(next_kernel, at + name.size());
    const size_t k = blob.find(want, at + name.size());
    if (k == std::string::npos || (end != std::string::npos && k > end)) return -1;
    const size_t p = k + want.size();
    if (p >= blob.size()) return -1;
    const unsigned char b = (unsigned char)blob[p];
    if (b < 0x80) return (int)b;
    if (b == 0xcc && p + 1 < blob.size()) return (int)(unsigned char)blob[p + 1];
    if (b == 0xcd && p + 2 < blob.size()) return (int)(((unsigned)(unsigned char)blob[p + 1] << 8) | (unsigned char)blob[p + 2]);
    return -1;
}

int jit_build(const JitPlan &plan, const std::string &cache_dir_in, JitBuild &out, bool cache_only) {
#if EU_REAL_BITS == 32
    const std::string fname = plan.key + "_f32.hsaco";
#else
    const std::string fname = plan.key + ".hsaco";
#endif
    {
        std::lock_guard<std::mutex> lk(g_mem_mutex);
        auto it = g_mem_cache.find(fname);
        if (it != g_mem_cache.end()) { out.code = it->second; out.from_cache = true; return EU_OK; }
    }
    const std::string user_dir = cache_dir_in.empty() ? default_cache_dir() : cache_dir_in;
    const std::string lib_dir = library_cache_dir();
    for (int which = 0; which < 2; which++) {
        const std::string &dir = which == 0 ? lib_dir : user_dir;
        if (dir.empty()) continue;
        if (which == 1 && !dir_is_private(dir)) continue;      /* (the directory next to the library is as trustworthy as the library) */
        if (read_file(dir + "/" + fname, out.code)) {
            out.from_cache = true;
            std::lock_guard<std::mutex> lk(g_mem_mutex);
            mem_cache_put(fname, out.code);
            return EU_OK;
        }
    }
    if (cache_only) { out.log = "not in any cache"; return EU_ERR_BUSY; }
    const auto t0 = std::chrono::steady_clock::now();
    int rc0 = hiprtc_compile(plan, nullptr, out.code, out.log);
    if (rc0 != EU_OK) return rc0;
    /* The fused shade kernel lives at a register cliff: at 128 VGPRs four of its waves share a SIMD, at 129 three (3d_room: 1.31 against 1.43 ms
     * per frame), and where the allocator lands depends on details of the scene's code.  When it lands just above, the kernel is compiled again
     * with launch bounds of four waves; that object is kept if it spills next to nothing (3d_room then: 128 VGPRs, 3 spilled, as fast as a
     * natural 127).  Kernels far above (the 4-D scenes' 150) are left alone: forced to 128 they spill 60-80 registers and lose 3 %. */
    if (plan.fused) {
        bool given = false;
        for (const std::string &f : plan.extra_flags) given = given || f.rfind("-DEU_FSHADE_WAVES=", 0) == 0 || f.rfind("-DEU_SHADE_WAVES=", 0) == 0;
        int lo = 128;      /* (tests: -DEU_JIT_CLIFF_LO=100 makes every scene's kernel count as "just above") */
        for (const std::string &f : plan.extra_flags) if (f.rfind("-DEU_JIT_CLIFF_LO=", 0) == 0) lo = atoi(f.c_str() + 18);
        const int v = given ? -1 : code_object_count(out.code, "eu_jit_fshade", ".vgpr_count");
        if (v > lo && v <= 136) {
            std::vector<char> code4; std::string log4;
            if (hiprtc_compile(plan, "-DEU_FSHADE_WAVES=4", code4, log4) == EU_OK) {
                const int v4 = code_object_count(code4, "eu_jit_fshade", ".vgpr_count"), s4 = code_object_count(code4, "eu_jit_fshade", ".vgpr_spill_count");
                if (v4 > 0 && v4 <= 128 && s4 >= 0 && s4 <= 8) {
                    out.code.swap(code4);
                    out.log += "\neu_jit_fshade: " + std::to_string(v) + " VGPRs -> compiled for four waves per SIMD (" + std::to_string(v4) + " VGPRs, " + std::to_string(s4) + " spilled)";
                }
            }
        }
    }
    out.compile_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    out.from_cache = false;
    if (!user_dir.empty()) {   /* keep it: atomically, so that a concurrent reader never sees half a file */
        mkdirs(user_dir);
        if (!dir_is_private(user_dir)) { std::lock_guard<std::mutex> lk(g_mem_mutex); mem_cache_put(fname, out.code); return EU_OK; }
        char tmpn[64];
        snprintf(tmpn, sizeof tmpn, ".tmp.%ld.%p", (long)getpid(), (void *)&out);
        const std::string tmp = user_dir + "/" + fname + tmpn, fin = user_dir + "/" + fname;
        FILE *f = fopen(tmp.c_str(), "wb");
        if (f) {
            const bool ok = fwrite(out.code.data(), 1, out.code.size(), f) == out.code.size();
            fclose(f);
            if (!ok || rename(tmp.c_str(), fin.c_str()) != 0) (void)unlink(tmp.c_str());
        }
    }
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    mem_cache_put(fname, out.code);
    return EU_OK;
}

/* a cached code object that the runtime refused to load: forget it (memory and the user's directory), so that the next jit_build compiles */
void jit_forget(const JitPlan &plan, const std::string &cache_dir_in) {
#if EU_REAL_BITS == 32
    const std::string fname = plan.key + "_f32.hsaco";
#else
    const std::string fname = plan.key + ".hsaco";
#endif
    {
        std::lock_guard<std::mutex> lk(g_mem_mutex);
        g_mem_cache.erase(fname);
        for (auto it = g_mem_order.begin(); it != g_mem_order.end(); ++it) if (*it == fname) { g_mem_order.erase(it); break; }
    }
    const std::string user_dir = cache_dir_in.empty() ? default_cache_dir() : cache_dir_in;
    if (!user_dir.empty() && dir_is_private(user_dir)) (void)unlink((user_dir + "/" + fname).c_str());
}

/* ------------------------------------------------------------------ asynchronous compilation (EU_SPECIALIZE_ASYNC) */
namespace {
struct JitWorker {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<JitJob>> queue;
    std::vector<std::weak_ptr<JitJob>> inflight;      /* queued or compiling, for jit_submit to join */
    std::thread thread;
    bool stop = false, started = false;
    void run() {
        for (;;) {
            std::shared_ptr<JitJob> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !queue.empty(); });
                if (stop) return;
                job = queue.front();
                queue.pop_front();
            }
            if (job->waiters.load() <= 0) { job->rc = EU_ERR_BUSY; job->done.store(true, std::memory_order_release); continue; }
            job->plan = jit_generate(*job->flat, job->flags, job->fused);
            job->rc = jit_build(job->plan, job->cache_dir, job->build);
            job->done.store(true, std::memory_order_release);
        }
    }
    ~JitWorker() {      /* process exit / library unload: no new work, and hiprtc is not torn down under the job in progress */
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            queue.clear();
        }
        cv.notify_all();
        if (started && thread.joinable()) thread.join();
    }
};
JitWorker &worker() { static JitWorker w; return w; }
}  // namespace

std::shared_ptr<JitJob> jit_submit(std::shared_ptr<const FlatScene> flat, const std::string &cache_dir, const std::string &flags, const std::string &key, bool fused) {
    JitWorker &w = worker();
    std::shared_ptr<JitJob> job;
    {
        std::lock_guard<std::mutex> lk(w.mu);
        /* the same scene, flags and cache directory already queued or compiling (the slots of a frame sequence, one renderer per device): join it */
        for (auto it = w.inflight.begin(); it != w.inflight.end();) {
            std::shared_ptr<JitJob> j = it->lock();
            if (!j || j->done.load(std::memory_order_acquire)) { it = w.inflight.erase(it); continue; }
            if (j->key == key && j->cache_dir == cache_dir) { j->waiters.fetch_add(1); return j; }
            ++it;
        }
        job = std::make_shared<JitJob>();
        job->flat = std::move(flat); job->cache_dir = cache_dir; job->flags = flags; job->key = key; job->fused = fused;
        job->waiters.store(1);
        if (!w.started) { w.thread = std::thread([&w] { w.run(); }); w.started = true; }
        w.queue.push_back(job);
        w.inflight.push_back(job);
    }
    w.cv.notify_one();
    return job;
}

}  // namespace euclider
