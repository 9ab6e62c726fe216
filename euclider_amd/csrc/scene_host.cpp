/*
 * scene_host.cpp -- JSON scene loader, universe constructors and flattener (host only).
 * See scene_host.hpp.  Citations: file:line under /root/reference/src/.
 */
#include "scene_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "eu_math.h"

/* Margins of the bounding spheres used for exact culling (trace_device.h, ray_misses_bound): the enlargement has to dwarf the
 * rounding of the device's test, so it depends on F.  f64: 1e-6 relative (rounding ~1e-15); f32: 1e-3 relative (rounding ~1e-6), and
 * the discriminant form of the test only within 10 radii instead of 10^4 (its cancellation error grows with the square of the distance). */
#if EU_REAL_BITS == 32
#define EU_BOUND_REL R(1.0e-3)
#define EU_BOUND_ABS R(1.0e-5)
#define EU_BOUND_FAR2 R(1.0e2)
#else
#define EU_BOUND_REL R(1.0e-6)
#define EU_BOUND_ABS R(1.0e-9)
#define EU_BOUND_FAR2 R(1.0e8)
#endif

namespace euclider {

/* ------------------------------------------------------------------ errors */
const char *ParserError::kind_name(Kind k) {
    switch (k) {
    case NoDeserializer: return "NoDeserializer";
    case SyntaxError: return "SyntaxError";
    case MissingType: return "MissingType";
    case InvalidConstructor: return "InvalidConstructor";
    case MissingField: return "MissingField";
    case TypeMismatch: return "TypeMismatch";
    default: return "CustomError";
    }
}
[[noreturn]] static void fail(ParserError::Kind k, const std::string &d) { throw ParserError{k, d}; }

/* ------------------------------------------------------------------ JSON */
namespace {
struct JsonReader {
    const char *s; size_t n, p = 0;
    [[noreturn]] void err(const char *what) { fail(ParserError::SyntaxError, std::string("Invalid JSON file. Please, check the syntax. (") + what + " at byte " + std::to_string(p) + ")"); }
    void ws() { while (p < n && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++; }
    bool lit(const char *w) { size_t l = strlen(w); if (p + l <= n && !memcmp(s + p, w, l)) { p += l; return true; } return false; }
    Json value(int depth) {
        if (depth > 256) err("nesting too deep");
        ws();
        if (p >= n) err("unexpected end");
        Json j;
        char c = s[p];
        if (c == '{') {
            p++; j.type = Json::Object; ws();
            if (p < n && s[p] == '}') { p++; return j; }
            for (;;) {
                ws();
                if (p >= n || s[p] != '"') err("expected string key");
                std::string k = string();
                ws();
                if (p >= n || s[p] != ':') err("expected ':'");
                p++;
                j.obj.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (p < n && s[p] == ',') { p++; continue; }
                if (p < n && s[p] == '}') { p++; return j; }
                err("expected ',' or '}'");
            }
        }
        if (c == '[') {
            p++; j.type = Json::Array; ws();
            if (p < n && s[p] == ']') { p++; return j; }
            for (;;) {
                j.arr.push_back(value(depth + 1));
                ws();
                if (p < n && s[p] == ',') { p++; continue; }
                if (p < n && s[p] == ']') { p++; return j; }
                err("expected ',' or ']'");
            }
        }
        if (c == '"') { j.type = Json::String; j.str = string(); return j; }
        if (lit("true")) { j.type = Json::Bool; j.b = true; return j; }
        if (lit("false")) { j.type = Json::Bool; j.b = false; return j; }
        if (lit("null")) { j.type = Json::Null; return j; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            size_t q = p;
            if (s[q] == '-') q++;
            if (q >= n || !(s[q] >= '0' && s[q] <= '9')) err("bad number");
            while (q < n && ((s[q] >= '0' && s[q] <= '9') || s[q] == '.' || s[q] == 'e' || s[q] == 'E' || s[q] == '+' || s[q] == '-')) q++;
            std::string tok(s + p, q - p);
            char *end = nullptr;
            j.num = strtod(tok.c_str(), &end);
            if (!end || *end) err("bad number");
            j.type = Json::Number;
            p = q;
            return j;
        }
        err("unexpected character");
    }
    std::string string() {
        std::string out;
        p++;
        while (p < n && s[p] != '"') {
            char c = s[p++];
            if (c == '\\') {
                if (p >= n) err("bad escape");
                char e = s[p++];
                switch (e) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case '/': out += '/'; break; case '\\': out += '\\'; break; case '"': out += '"'; break;
                case 'u': {
                    if (p + 4 > n) err("bad \\u escape");
                    unsigned cp = (unsigned)strtoul(std::string(s + p, 4).c_str(), nullptr, 16);
                    p += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: err("bad escape");
                }
            } else out += c;
        }
        if (p >= n) err("unterminated string");
        p++;
        return out;
    }
};
}  // namespace

Json Json::parse(const char *text, size_t len) {
    JsonReader r{text, len};
    Json j = r.value(0);
    r.ws();
    if (r.p != len) r.err("trailing characters");
    return j;
}
const Json *Json::get(const std::string &key) const {
    for (auto &kv : obj) if (kv.first == key) return &kv.second;
    return nullptr;
}
std::string Json::brief() const {
    switch (type) {
    case Null: return "null";
    case Bool: return b ? "true" : "false";
    case Number: { char buf[64]; snprintf(buf, sizeof buf, "%g", num); return buf; }
    case String: return "\"" + str + "\"";
    case Array: return "[...]";
    default: { std::string s = "{"; for (auto &kv : obj) { s += "\"" + kv.first + "\": ..."; break; } return s + "}"; }
    }
}

/* ------------------------------------------------------------------ vectors (nalgebra 0.8.2; x -> w order) */
static real v_dot(int D, const real *a, const real *b) { real s = a[0] * b[0]; for (int i = 1; i < D; i++) s = s + a[i] * b[i]; return s; }
static real v_nsq(int D, const real *a) { return v_dot(D, a, a); }
static void v_normalize(int D, const real *a, real *o) { real n = sqrt(v_nsq(D, a)); for (int i = 0; i < D; i++) o[i] = a[i] / n; }

/* ------------------------------------------------------------------ shapes */
ShapePtr VoidShape_new(int dim) { auto s = std::make_shared<Shape>(); s->kind = Shape::VoidShape; s->dim = dim; return s; }

ShapePtr Sphere_new(int dim, const real *center, real radius) {           /* shape.rs:643-649 */
    auto s = std::make_shared<Shape>(); s->kind = Shape::Sphere; s->dim = dim;
    for (int i = 0; i < dim; i++) s->a[i] = center[i];
    s->r = radius;
    return s;
}
ShapePtr Hyperplane_new(int dim, const real *normal, real constant) {     /* shape.rs:750-759 */
    if (!(v_nsq(dim, normal) > R(0.0))) fail(ParserError::CustomError, "Cannot have a normal with length of 0.");
    auto s = std::make_shared<Shape>(); s->kind = Shape::Hyperplane; s->dim = dim;
    for (int i = 0; i < dim; i++) s->a[i] = normal[i];
    s->r = constant;
    return s;
}
ShapePtr Hyperplane_new_with_point(int dim, const real *normal, const real *point) {   /* shape.rs:761-766 */
    real constant = -v_dot(dim, normal, point);
    return Hyperplane_new(dim, normal, constant);
}
ShapePtr Hyperplane_new_with_vectors(const real *a, const real *b, const real *point) {   /* shape.rs:768-776 */
    real n[MAXD] = {0, 0, 0, 0};
    n[0] = a[1] * b[2] - a[2] * b[1];
    n[1] = a[2] * b[0] - a[0] * b[2];
    n[2] = a[0] * b[1] - a[1] * b[0];
    return Hyperplane_new_with_point(3, n, point);
}
ShapePtr HalfSpace_new(const ShapePtr &plane, real sign) {                  /* shape.rs:828-835 */
    if (!plane || plane->kind != Shape::Hyperplane)
        fail(ParserError::CustomError, std::string("Invalid type, expected a `Hyperplane") + (plane && plane->dim == 4 ? "4" : "3") + "`.");
    auto s = std::make_shared<Shape>(); s->kind = Shape::HalfSpace; s->dim = plane->dim;
    for (int i = 0; i < plane->dim; i++) s->a[i] = plane->a[i];
    s->r = plane->r;
    s->signum = sign / fabs(sign);
    return s;
}
ShapePtr HalfSpace_new_with_point(const ShapePtr &plane, const real *point) {   /* shape.rs:837-841 */
    if (!plane || plane->kind != Shape::Hyperplane)
        fail(ParserError::CustomError, std::string("Invalid type, expected a `Hyperplane") + (plane && plane->dim == 4 ? "4" : "3") + "`.");
    real identifier = v_dot(plane->dim, plane->a, point) + plane->r;
    return HalfSpace_new(plane, identifier);
}
ShapePtr ComposableShape_of(const std::vector<ShapePtr> &shapes, SetOperation op) {   /* shape.rs:523-545: left fold */
    if (shapes.size() < 2) fail(ParserError::CustomError, "2 or more `Shape`s are needed to construct a `ComposableShape`.");
    ShapePtr result = shapes[0];
    for (size_t i = 1; i < shapes.size(); i++) {
        auto c = std::make_shared<Shape>(); c->kind = Shape::ComposableShape; c->dim = shapes[0]->dim;
        c->operation = op; c->sa = result; c->sb = shapes[i];
        result = c;
    }
    return result;
}
static ShapePtr box_of_halfspaces(int D, const real *center, const real *abc) {
    real half[MAXD];
    for (int i = 0; i < D; i++) half[i] = abc[i] / R(2.0);
    real axis[MAXD][MAXD] = {{0}};
    for (int i = 0; i < D; i++) axis[i][i] = R(1.0);
    std::vector<ShapePtr> shapes;
    for (int ax = 0; ax < D; ax++) for (int neg = 0; neg < 2; neg++) {
        real off[MAXD], pt[MAXD];
        for (int i = 0; i < D; i++) off[i] = axis[ax][i] * half[i];            /* x * half_abc is component-wise */
        if (neg) for (int i = 0; i < D; i++) off[i] = -off[i];
        for (int i = 0; i < D; i++) pt[i] = center[i] + off[i];                /* na::translate(&v, &center) */
        ShapePtr plane;
        if (D == 3) {                                                          /* d3/entity/shape.rs:22-63 */
            const real *va = (ax == 0) ? axis[1] : axis[0];
            const real *vb = (ax == 2) ? axis[1] : axis[2];
            plane = Hyperplane_new_with_vectors(va, vb, pt);
        } else {                                                               /* d4/entity/shape.rs:25-72 */
            real nn[MAXD];
            v_normalize(D, axis[ax], nn);
            plane = Hyperplane_new_with_point(D, nn, pt);
        }
        shapes.push_back(HalfSpace_new_with_point(plane, center));
    }
    return ComposableShape_of(shapes, SetOperation::Intersection);
}
ShapePtr HalfSpace_cuboid(const real *center, const real *abc) { return box_of_halfspaces(3, center, abc); }
ShapePtr HalfSpace_hypercuboid(const real *center, const real *abcd) { return box_of_halfspaces(4, center, abcd); }

ShapePtr Cylinder_new(int dim, const real *center, const real *direction, real radius) {   /* shape.rs:893-904 */
    if (!(v_nsq(dim, direction) > R(0.0))) fail(ParserError::CustomError, "Cannot have a direction with length of 0.");
    if (!(radius > R(0.0))) fail(ParserError::CustomError, "The radius must be positive.");
    auto s = std::make_shared<Shape>(); s->kind = Shape::Cylinder; s->dim = dim;
    for (int i = 0; i < dim; i++) s->a[i] = center[i];
    v_normalize(dim, direction, s->b);
    s->r = radius;
    return s;
}
ShapePtr Cylinder_new_with_height(int dim, const real *center, const real *direction, real radius, real height) {   /* shape.rs:906-927 */
    real nd[MAXD], pt[MAXD];
    v_normalize(dim, direction, nd);
    real half_height = height / (R(1.0) + R(1.0));
    std::vector<ShapePtr> shapes;
    shapes.push_back(Cylinder_new(dim, center, direction, radius));
    for (int i = 0; i < dim; i++) pt[i] = center[i] + nd[i] * half_height;
    shapes.push_back(HalfSpace_new_with_point(Hyperplane_new_with_point(dim, nd, pt), center));
    real neg_half = -half_height;
    for (int i = 0; i < dim; i++) pt[i] = center[i] + nd[i] * neg_half;
    shapes.push_back(HalfSpace_new_with_point(Hyperplane_new_with_point(dim, nd, pt), center));
    return ComposableShape_of(shapes, SetOperation::Intersection);
}

/* ------------------------------------------------------------------ misc constructors */
eu_camera default_camera(int dim, const real *loc) {   /* d3/entity/camera.rs:42-52, d4/entity/camera.rs:47-58 */
    eu_camera c;
    memset(&c, 0, sizeof c);
    c.dim = dim;
    if (loc) for (int i = 0; i < dim; i++) c.location[i] = loc[i];
    c.forward[0] = R(1.0);
    c.up[2] = R(1.0);
    c.left[1] = R(1.0);
    c.fov_deg = 90;
    c.max_depth = 10;
    return c;
}

/* palette 0.2.1 Hsv -> Rgb with RgbHue::to_positive_degrees (UNVERIFIED third-party semantics) */
void rgba_from_hsva(real hue, real saturation, real value, real alpha, real *out) {
    real deg = hue;
    if (fabs(deg) < R(1.0e9)) {
        while (deg >= R(360.0)) deg = deg - R(360.0);
        while (deg < R(0.0)) deg = deg + R(360.0);
    }
    real c = value * saturation;
    real h = deg / R(60.0);
    real x = c * (R(1.0) - fabs(fmod(h, R(2.0)) - R(1.0)));
    real m = value - c;
    real r, g, b;
    if (h >= R(0.0) && h < R(1.0)) { r = c; g = x; b = R(0.0); }
    else if (h >= R(1.0) && h < R(2.0)) { r = x; g = c; b = R(0.0); }
    else if (h >= R(2.0) && h < R(3.0)) { r = R(0.0); g = c; b = x; }
    else if (h >= R(3.0) && h < R(4.0)) { r = R(0.0); g = x; b = c; }
    else if (h >= R(4.0) && h < R(5.0)) { r = x; g = R(0.0); b = c; }
    else { r = c; g = R(0.0); b = x; }
    out[0] = r + m; out[1] = g + m; out[2] = b + m; out[3] = alpha;
}

/* deterministic stand-in for textures the reference repository does not ship
 * (resources/universe_dim.jpg, .MISSING_LARGE_BLOBS) */
void procedural_uv_grid(uint32_t w, uint32_t h, std::vector<uint8_t> &rgba) {
    rgba.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) {
        uint8_t *p = &rgba[((size_t)y * w + x) * 4];
        if (x % 64 == 0 || y % 64 == 0) { p[0] = 255; p[1] = 255; p[2] = 255; p[3] = 255; continue; }
        p[0] = (uint8_t)(x * 255u / (w > 1 ? w - 1 : 1));
        p[1] = (uint8_t)(y * 255u / (h > 1 ? h - 1 : 1));
        p[2] = (uint8_t)(64u + 128u * (((x / 32) + (y / 32)) % 2));
        p[3] = 255;
    }
}

/* ------------------------------------------------------------------ expressions (meval 0.1.0 subset) */
namespace {
const char *FN_NAMES[EU_FN_COUNT] = {"sqrt", "abs", "floor", "ceil", "min", "max", "sin", "cos", "tan", "asin", "acos", "atan", "atan2", "signum"};
const int FN_ARITY[EU_FN_COUNT] = {1, 1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 2, 1};

/* Recursive descent emitting RPN directly.  Precedence (meval's shunting yard, UNVERIFIED):
 * + - (left) < * / % (left) < unary +- < ^ (right). */
struct ExprCompiler {
    const std::string &s; size_t p = 0; std::vector<Expr::Tok> &out; bool ok = true;
    void ws() { while (p < s.size() && isspace((unsigned char)s[p])) p++; }
    void emit(uint32_t op, uint32_t arg = 0, eu_f64 k = 0.0, const std::string &v = "") { out.push_back({op, arg, k, v}); }
    void atom() {
        ws();
        if (p >= s.size()) { ok = false; return; }
        char c = s[p];
        if (c == '(') { p++; expr(); ws(); if (p < s.size() && s[p] == ')') p++; else ok = false; return; }
        if (isdigit((unsigned char)c) || c == '.') {
            const char *b = s.c_str() + p; char *e = nullptr;
            eu_f64 v = strtod(b, &e);
            if (e == b) { ok = false; return; }
            p += (size_t)(e - b);
            emit(EU_RPN_CONST, 0, v);
            return;
        }
        if (isalpha((unsigned char)c) || c == '_') {
            std::string name;
            while (p < s.size() && (isalnum((unsigned char)s[p]) || s[p] == '_')) name += s[p++];
            ws();
            if (p < s.size() && s[p] == '(') {
                int fn = -1;
                for (int i = 0; i < (int)EU_FN_COUNT; i++) if (name == FN_NAMES[i]) fn = i;
                if (fn < 0) { ok = false; return; }
                p++;
                expr(); ws();
                if (FN_ARITY[fn] == 2) { if (p < s.size() && s[p] == ',') { p++; expr(); ws(); } else { ok = false; return; } }
                if (p < s.size() && s[p] == ')') p++; else { ok = false; return; }
                emit(EU_RPN_FN, (uint32_t)fn);
                return;
            }
            if (name == "pi") { emit(EU_RPN_CONST, 0, 3.14159265358979323846264338327950288); return; }
            if (name == "e") { emit(EU_RPN_CONST, 0, 2.71828182845904523536028747135266250); return; }
            emit(EU_RPN_VAR, 0, R(0.0), name);
            return;
        }
        ok = false;
    }
    void power() { atom(); ws(); if (ok && p < s.size() && s[p] == '^') { p++; unary(); emit(EU_RPN_POW); } }
    void unary() {
        ws();
        if (p < s.size() && s[p] == '-') { p++; unary(); emit(EU_RPN_NEG); return; }
        if (p < s.size() && s[p] == '+') { p++; unary(); return; }
        power();
    }
    void term() {
        unary();
        for (;;) {
            ws();
            if (!ok || p >= s.size() || (s[p] != '*' && s[p] != '/' && s[p] != '%')) return;
            char c = s[p++];
            unary();
            emit(c == '*' ? EU_RPN_MUL : c == '/' ? EU_RPN_DIV : EU_RPN_REM);
        }
    }
    void expr() {
        term();
        for (;;) {
            ws();
            if (!ok || p >= s.size() || (s[p] != '+' && s[p] != '-')) return;
            char c = s[p++];
            term();
            emit(c == '+' ? EU_RPN_ADD : EU_RPN_SUB);
        }
    }
};
}  // namespace

Expr Expr::from_str(const std::string &src) {
    Expr e; e.source = src;
    ExprCompiler c{src, 0, e.rpn};
    c.expr(); c.ws();
    if (!c.ok || c.p != src.size() || e.rpn.empty())
        fail(ParserError::CustomError, "Invalid component transformation expression `" + src + "`.");
    return e;
}
int Expr::stack_depth() const {
    int d = 0, mx = 0;
    for (auto &t : rpn) {
        switch (t.op) {
        case EU_RPN_CONST: case EU_RPN_VAR: d++; break;
        case EU_RPN_NEG: break;
        case EU_RPN_FN: d -= FN_ARITY[t.arg] - 1; break;
        default: d--; break;
        }
        if (d > mx) mx = d;
    }
    return mx;
}

/* ------------------------------------------------------------------ parser (scene.rs:554-1478) */
namespace {
enum class T { F, U8, U32, Str, Point, Vector, Rgba, Entity, Shape, Material, Surface, SetOperation, Expr,
               LinearTransformation, UVFn, Texture, MappedTexture, ReflectionRatio, ReflectionDirection,
               ThresholdDirection, SurfaceColor, BlendFunction, Camera, Environment };
struct Ty { T t; int dim; bool vec; };
static Ty ty(T t, int dim = 0, bool vec = false) { return Ty{t, dim, vec}; }
static std::string ty_name(const Ty &x) {
    static const char *names[] = {"F", "u8", "u32", "&str", "Point", "Vector", "Rgba<F>", "Entity", "Shape", "Material", "Surface",
        "SetOperation", "ComponentTransformationExpr", "LinearTransformation", "UVFn", "Texture", "MappedTexture",
        "ReflectionRatioProvider", "ReflectionDirectionProvider", "ThresholdDirectionProvider", "SurfaceColorProvider",
        "BlendFunction", "Camera", "Environment"};
    std::string s = names[(int)x.t];
    if (x.dim) s += std::to_string(x.dim);
    if (x.vec) s = "Vec<" + s + ">";
    return s;
}
struct Value {
    real num = R(0.0);
    std::string str;
    std::array<real, 4> vec{{0, 0, 0, 0}};
    std::shared_ptr<void> obj;
    std::vector<Value> list;
};
struct Field { const char *name; Ty type; };
struct Parser_;
struct Ctor {
    std::vector<Field> fields;
    Ty product;
    std::function<Value(Parser_ &, std::vector<Value> &)> build;
};
template <class X> std::shared_ptr<X> as(const Value &v) { return std::static_pointer_cast<X>(v.obj); }
template <class X> Value wrap(std::shared_ptr<X> p) { Value v; v.obj = std::move(p); return v; }
}  // namespace

struct Parser::Impl {};

namespace {
struct Parser_ {
    eu_load_opts opts{};
    uint32_t substituted = 0;
    std::map<std::string, Ctor> reg;

    void add(std::initializer_list<const char *> names, std::vector<Field> fields, Ty product,
             std::function<Value(Parser_ &, std::vector<Value> &)> build) {
        for (auto n : names) reg[n] = Ctor{fields, product, build};
    }

    std::shared_ptr<Texture> load_texture(const std::string &path, uint32_t kind) {
        auto t = std::make_shared<Texture>();
        t->kind = kind; t->path = path;
        t->rgba = std::make_shared<std::vector<uint8_t>>();
        uint32_t w = 0, h = 0; uint8_t *px = nullptr;
        int rc = -1;
        if (opts.load_texture) rc = opts.load_texture(opts.user, path.c_str(), &w, &h, &px);
        if (rc == 0 && px && w && h) {
            t->w = w; t->h = h;
            t->rgba->assign(px, px + (size_t)w * h * 4);
            free(px);
        } else {
            /* image::open would fail: "Could not load texture" (scene.rs:1053-1068).  The
             * benchmark scenes reference a file the reference repository does not ship, so
             * the loader substitutes the documented procedural grid and counts it. */
            if (px) free(px);
            t->w = 1024; t->h = 512;
            procedural_uv_grid(t->w, t->h, *t->rgba);
            substituted++;
        }
        return t;
    }

    Value field(const Ty &type, const Json &j, const Json &parent) {
        if (type.vec) {
            if (j.type != Json::Array) fail(ParserError::TypeMismatch, "Expected an array for `" + ty_name(type) + "`, could not parse from `" + j.brief() + "`.");
            Value v;
            Ty inner = type; inner.vec = false;
            for (auto &e : j.arr) v.list.push_back(field(inner, e, parent));
            return v;
        }
        switch (type.t) {
        case T::F: {
            if (j.type != Json::Number) fail(ParserError::TypeMismatch, "Expected `floating point number`, could not parse from `" + j.brief() + "`.");
            Value v; v.num = j.num; return v;
        }
        case T::U8: case T::U32: {
            real lim = type.t == T::U8 ? R(255.0) : R(4294967295.0);
            if (j.type != Json::Number || !(j.num >= R(0.0) && j.num <= lim) || j.num != floor(j.num))
                fail(ParserError::TypeMismatch, std::string("Expected `") + (type.t == T::U8 ? "8-bit unsigned integer" : "32-bit unsigned integer") + "`, could not parse from `" + j.brief() + "`.");
            Value v; v.num = j.num; return v;
        }
        case T::Str: {
            if (j.type != Json::String) fail(ParserError::TypeMismatch, "Expected `string`, could not parse from `" + j.brief() + "`.");
            Value v; v.str = j.str; return v;
        }
        default: return construct(type, j);
        }
    }

    /* deserialize_constructor + deserialize (scene.rs:1430-1464) */
    Value construct(const Ty &expected, const Json &j) {
        if (j.type != Json::Object || j.obj.size() != 1)
            fail(ParserError::InvalidConstructor, "A constructor must be an object containing a single key pointing to either an object or an array.");
        const std::string &key = j.obj[0].first;
        const Json &data = j.obj[0].second;
        auto it = reg.find(key);
        if (it == reg.end()) fail(ParserError::NoDeserializer, "No deserializer registered for key `" + key + "`.");
        const Ctor &c = it->second;
        if (c.product.t != expected.t || c.product.dim != expected.dim)
            fail(ParserError::TypeMismatch, "The constructor used (`" + key + "`) has an incorrect type for this field. (expected " +
                                            ty_name(expected) + ", it produces " + ty_name(c.product) + ")");
        std::vector<Value> args;
        if (data.type == Json::Object) {
            for (auto &f : c.fields) {
                const Json *v = data.get(f.name);
                if (!v) fail(ParserError::MissingField, "Missing field of type " + ty_name(f.type) + " with key " + f.name + " in `" + key + "`.");
                args.push_back(field(f.type, *v, j));
            }
        } else if (data.type == Json::Array) {
            size_t i = 0;
            for (auto &f : c.fields) {
                if (i >= data.arr.size())
                    fail(ParserError::MissingField, "Missing field of type " + ty_name(f.type) + " in `" + key + "`. To fix this, add the field at the end of the array.");
                args.push_back(field(f.type, data.arr[i++], j));
            }
        } else {
            fail(ParserError::InvalidConstructor, "The constructor data may only be an array or an object, received " + data.brief() + " instead.");
        }
        return c.build(*this, args);
    }

    void build_registry();
};

static const real *vp(const Value &v) { return v.vec.data(); }

void Parser_::build_registry() {
    for (int d = 3; d <= 4; d++) {
        std::vector<Field> comps = {{"x", ty(T::F)}, {"y", ty(T::F)}, {"z", ty(T::F)}};
        if (d == 4) comps.push_back({"w", ty(T::F)});
        auto mkvec = [d](Parser_ &, std::vector<Value> &a) { Value v; for (int i = 0; i < d; i++) v.vec[i] = a[i].num; return v; };
        std::string P = "Point" + std::to_string(d), V = "Vector" + std::to_string(d);
        reg[P] = reg[P + "::new"] = Ctor{comps, ty(T::Point, d), mkvec};                      /* scene.rs:620-633 */
        reg[V] = reg[V + "::new"] = Ctor{comps, ty(T::Vector, d), mkvec};                     /* scene.rs:634-647 */
    }
    std::vector<Field> rgba_f = {{"r", ty(T::F)}, {"g", ty(T::F)}, {"b", ty(T::F)}, {"a", ty(T::F)}};
    add({"Rgba", "Rgba::new"}, rgba_f, ty(T::Rgba), [](Parser_ &, std::vector<Value> &a) { Value v; for (int i = 0; i < 4; i++) v.vec[i] = a[i].num; return v; });
    add({"Rgba::new_u8"}, {{"r", ty(T::U8)}, {"g", ty(T::U8)}, {"b", ty(T::U8)}, {"a", ty(T::U8)}}, ty(T::Rgba),
        [](Parser_ &, std::vector<Value> &a) { Value v; for (int i = 0; i < 4; i++) v.vec[i] = a[i].num / R(255.0); return v; });
    add({"Rgba::from_hsva"}, {{"hue", ty(T::F)}, {"saturation", ty(T::F)}, {"value", ty(T::F)}, {"alpha", ty(T::F)}}, ty(T::Rgba),
        [](Parser_ &, std::vector<Value> &a) { Value v; rgba_from_hsva(a[0].num, a[1].num, a[2].num, a[3].num, v.vec.data()); return v; });
    add({"SetOperation", "SetOperation::new"}, {{"name", ty(T::Str)}}, ty(T::SetOperation), [](Parser_ &, std::vector<Value> &a) {   /* scene.rs:774-787 */
        Value v;
        if (a[0].str == "Union") v.num = 0; else if (a[0].str == "Intersection") v.num = 1;
        else if (a[0].str == "Complement") v.num = 2; else if (a[0].str == "SymmetricDifference") v.num = 3;
        else fail(ParserError::CustomError, "Invalid `SetOperation`: \"" + a[0].str + "\"");
        return v;
    });
    add({"ComponentTransformationExpr", "ComponentTransformationExpr::new"}, {{"expression", ty(T::Str)}, {"inverse_expression", ty(T::Str)}}, ty(T::Expr),
        [](Parser_ &, std::vector<Value> &a) {                                                /* scene.rs:961-989 */
            auto e = std::make_shared<ComponentTransformationExpr>();
            e->expression = Expr::from_str(a[0].str);
            e->inverse_expression = Expr::from_str(a[1].str);
            return wrap(e);
        });
    add({"uv_sphere_3"}, {{"center", ty(T::Point, 3)}}, ty(T::UVFn, 3), [](Parser_ &, std::vector<Value> &a) {   /* scene.rs:1037-1042 */
        auto u = std::make_shared<UVFn>(); u->dim = 3; for (int i = 0; i < 3; i++) u->center[i] = a[0].vec[i]; return wrap(u);
    });
    add({"uv_derank_4"}, {{"uvfn", ty(T::UVFn, 3)}}, ty(T::UVFn, 4), [](Parser_ &, std::vector<Value> &a) {      /* scene.rs:1044-1049 */
        auto u = std::make_shared<UVFn>(*as<UVFn>(a[0])); u->dim = 4; return wrap(u);
    });
    add({"texture_image_nearest_neighbor"}, {{"path", ty(T::Str)}}, ty(T::Texture), [](Parser_ &p, std::vector<Value> &a) { return wrap(p.load_texture(a[0].str, EU_TEX_NEAREST)); });
    add({"texture_image_linear"}, {{"path", ty(T::Str)}}, ty(T::Texture), [](Parser_ &p, std::vector<Value> &a) { return wrap(p.load_texture(a[0].str, EU_TEX_LINEAR)); });
    add({"blend_function_ratio"}, {{"ratio", ty(T::F)}}, ty(T::BlendFunction), [](Parser_ &, std::vector<Value> &a) {
        auto b = std::make_shared<BlendFunction>(); b->fn = EU_BL_RATIO; b->ratio = a[0].num; return wrap(b);
    });
    static const char *blend_names[] = {"over", "inside", "outside", "atop", "xor", "plus", "multiply", "screen", "overlay", "darken",
                                        "lighten", "dodge", "burn", "hard_light", "soft_light", "difference", "exclusion"};
    for (uint32_t i = 0; i < 17; i++) {                                                       /* scene.rs:1131-1158 */
        std::string name = std::string("blend_function_") + blend_names[i];
        reg[name] = Ctor{{}, ty(T::BlendFunction), [i](Parser_ &, std::vector<Value> &) { auto b = std::make_shared<BlendFunction>(); b->fn = i; return wrap(b); }};
    }

    for (int d = 3; d <= 4; d++) {
        const std::string n = std::to_string(d);
        Ty P = ty(T::Point, d), V = ty(T::Vector, d), SH = ty(T::Shape, d), MAT = ty(T::Material, d), SURF = ty(T::Surface, d), ENT = ty(T::Entity, d);
        Ty RR = ty(T::ReflectionRatio, d), RD = ty(T::ReflectionDirection, d), TD = ty(T::ThresholdDirection, d), SC = ty(T::SurfaceColor, d);
        auto N = [&](const char *fmt) { std::string s = fmt; size_t k; while ((k = s.find('#')) != std::string::npos) s.replace(k, 1, n); return s; };
        auto addn = [&](std::initializer_list<const char *> names, std::vector<Field> f, Ty prod, std::function<Value(Parser_ &, std::vector<Value> &)> b) {
            for (auto nm : names) reg[N(nm)] = Ctor{f, prod, b};
        };
        /* entities, scene.rs:672-738 */
        addn({"Void#", "Void#::new"}, {{"material", MAT}}, ENT, [](Parser_ &, std::vector<Value> &a) {
            auto e = std::make_shared<Entity>(); e->material = as<Material>(a[0]); e->shape = VoidShape_new(e->material->dim); return wrap(e); });
        addn({"Void#::new_with_vacuum"}, {}, ENT, [d](Parser_ &, std::vector<Value> &) {
            auto e = std::make_shared<Entity>(); e->material = std::make_shared<Material>(); e->material->dim = d; e->shape = VoidShape_new(d); return wrap(e); });
        addn({"Entity#Impl", "Entity#Impl::new", "Entity#Impl::new_with_surface"}, {{"shape", SH}, {"material", MAT}, {"surface", SURF}}, ENT,
             [](Parser_ &, std::vector<Value> &a) { auto e = std::make_shared<Entity>(); e->shape = as<Shape>(a[0]); e->material = as<Material>(a[1]); e->surface = as<ComposableSurface>(a[2]); return wrap(e); });
        addn({"Entity#Impl::new_without_surface"}, {{"shape", SH}, {"material", MAT}}, ENT,
             [](Parser_ &, std::vector<Value> &a) { auto e = std::make_shared<Entity>(); e->shape = as<Shape>(a[0]); e->material = as<Material>(a[1]); return wrap(e); });
        /* shapes, scene.rs:742-943 */
        addn({"VoidShape#", "VoidShape#::new"}, {}, SH, [d](Parser_ &, std::vector<Value> &) { return wrap(VoidShape_new(d)); });
        addn({"ComposableShape#", "ComposableShape#::new", "ComposableShape#::of"}, {{"shapes", ty(T::Shape, d, true)}, {"operation", ty(T::SetOperation)}}, SH,
             [](Parser_ &, std::vector<Value> &a) { std::vector<ShapePtr> s; for (auto &v : a[0].list) s.push_back(as<Shape>(v)); return wrap(ComposableShape_of(s, (SetOperation)(int)a[1].num)); });
        addn({"Sphere#", "Sphere#::new"}, {{"center", P}, {"radius", ty(T::F)}}, SH, [d](Parser_ &, std::vector<Value> &a) { return wrap(Sphere_new(d, vp(a[0]), a[1].num)); });
        addn({"Hyperplane#", "Hyperplane#::new"}, {{"normal", V}, {"constant", ty(T::F)}}, SH, [d](Parser_ &, std::vector<Value> &a) { return wrap(Hyperplane_new(d, vp(a[0]), a[1].num)); });
        addn({"Hyperplane#::new_with_point"}, {{"normal", V}, {"point", P}}, SH, [d](Parser_ &, std::vector<Value> &a) { return wrap(Hyperplane_new_with_point(d, vp(a[0]), vp(a[1]))); });
        if (d == 3)
            addn({"Hyperplane3::new_with_vectors"}, {{"first", V}, {"second", V}, {"point", P}}, SH, [](Parser_ &, std::vector<Value> &a) { return wrap(Hyperplane_new_with_vectors(vp(a[0]), vp(a[1]), vp(a[2]))); });
        addn({"HalfSpace#", "HalfSpace#::new"}, {{"plane", SH}, {"sign", ty(T::F)}}, SH, [](Parser_ &, std::vector<Value> &a) { return wrap(HalfSpace_new(as<Shape>(a[0]), a[1].num)); });
        addn({"HalfSpace#::new_with_point"}, {{"plane", SH}, {"point", P}}, SH, [](Parser_ &, std::vector<Value> &a) { return wrap(HalfSpace_new_with_point(as<Shape>(a[0]), vp(a[1]))); });
        if (d == 3) addn({"HalfSpace3::cuboid"}, {{"center", P}, {"dimensions", V}}, SH, [](Parser_ &, std::vector<Value> &a) { return wrap(HalfSpace_cuboid(vp(a[0]), vp(a[1]))); });
        else addn({"HalfSpace4::hypercuboid"}, {{"center", P}, {"dimensions", V}}, SH, [](Parser_ &, std::vector<Value> &a) { return wrap(HalfSpace_hypercuboid(vp(a[0]), vp(a[1]))); });
        addn({"Cylinder#", "Cylinder#::new"}, {{"center", P}, {"direction", V}, {"radius", ty(T::F)}}, SH,
             [d](Parser_ &, std::vector<Value> &a) { return wrap(Cylinder_new(d, vp(a[0]), vp(a[1]), a[2].num)); });
        addn({"Cylinder#::new_with_height"}, {{"center", P}, {"direction", V}, {"radius", ty(T::F)}, {"height", ty(T::F)}}, SH,
             [d](Parser_ &, std::vector<Value> &a) { return wrap(Cylinder_new_with_height(d, vp(a[0]), vp(a[1]), a[2].num, a[3].num)); });
        /* materials, scene.rs:947-1033 */
        addn({"Vacuum#", "Vacuum#::new"}, {}, MAT, [d](Parser_ &, std::vector<Value> &) { auto m = std::make_shared<Material>(); m->dim = d; return wrap(m); });
        addn({"ComponentTransformation#", "ComponentTransformation#::new"}, {{"expressions", ty(T::Expr, 0, true)}}, ty(T::LinearTransformation, d),
             [](Parser_ &, std::vector<Value> &a) { auto t = std::make_shared<ComponentTransformation>(); for (auto &v : a[0].list) t->expressions.push_back(*as<ComponentTransformationExpr>(v)); return wrap(t); });
        addn({"LinearSpace#", "LinearSpace#::new"}, {{"legend", ty(T::Str)}, {"transformations", ty(T::LinearTransformation, d, true)}}, MAT,
             [d](Parser_ &, std::vector<Value> &a) {
                 auto m = std::make_shared<Material>(); m->kind = Material::LinearSpace; m->dim = d; m->legend = a[0].str;
                 for (auto &v : a[1].list) m->transformations.push_back(as<ComponentTransformation>(v));
                 return wrap(m);
             });
        /* textures + surfaces, scene.rs:1075-1335 */
        addn({"MappedTextureImpl#", "MappedTextureImpl#::new"}, {{"uvfn", ty(T::UVFn, d)}, {"texture", ty(T::Texture)}}, ty(T::MappedTexture, d),
             [d](Parser_ &, std::vector<Value> &a) { auto m = std::make_shared<MappedTexture>(); m->dim = d; m->uvfn = as<UVFn>(a[0]); m->texture = as<Texture>(a[1]); return wrap(m); });
        addn({"ComposableSurface#", "ComposableSurface#::new"},
             {{"reflection_ratio", RR}, {"reflection_direction", RD}, {"threshold_direction", TD}, {"surface_color", SC}}, SURF,
             [](Parser_ &, std::vector<Value> &a) {
                 auto s = std::make_shared<ComposableSurface>();
                 s->reflection_ratio = as<ReflectionRatio>(a[0]); s->reflection_direction = as<ReflectionDirection>(a[1]);
                 s->threshold_direction = as<ThresholdDirection>(a[2]); s->surface_color = as<SurfaceColor>(a[3]);
                 return wrap(s);
             });
        addn({"surface_color_blend_#"}, {{"source", SC}, {"destination", SC}, {"blend_function", ty(T::BlendFunction)}}, SC,
             [d](Parser_ &, std::vector<Value> &a) { auto c = std::make_shared<SurfaceColor>(); c->kind = EU_COL_BLEND; c->dim = d; c->source = as<SurfaceColor>(a[0]); c->destination = as<SurfaceColor>(a[1]); c->blend = as<BlendFunction>(a[2]); return wrap(c); });
        addn({"surface_color_illumination_global_#"}, {{"light_color", ty(T::Rgba)}, {"dark_color", ty(T::Rgba)}}, SC,
             [d](Parser_ &, std::vector<Value> &a) { auto c = std::make_shared<SurfaceColor>(); c->kind = EU_COL_ILLUM_GLOBAL; c->dim = d; for (int i = 0; i < 4; i++) { c->c0[i] = a[0].vec[i]; c->c1[i] = a[1].vec[i]; } return wrap(c); });
        addn({"surface_color_illumination_directional_#"}, {{"direction", V}, {"light_color", ty(T::Rgba)}, {"dark_color", ty(T::Rgba)}}, SC,
             [d](Parser_ &, std::vector<Value> &a) { auto c = std::make_shared<SurfaceColor>(); c->kind = EU_COL_ILLUM_DIR; c->dim = d; for (int i = 0; i < 4; i++) { c->v[i] = a[0].vec[i]; c->c0[i] = a[1].vec[i]; c->c1[i] = a[2].vec[i]; } return wrap(c); });
        if (d == 3) {
            addn({"surface_color_perlin_hue_seed_3"}, {{"seed", ty(T::U32)}, {"size", ty(T::F)}, {"speed", ty(T::F)}}, SC,
                 [](Parser_ &, std::vector<Value> &a) { auto c = std::make_shared<SurfaceColor>(); c->kind = EU_COL_PERLIN; c->seed = (uint32_t)a[0].num; c->v[0] = a[1].num; c->v[1] = a[2].num; return wrap(c); });
            addn({"surface_color_perlin_hue_random_3"}, {{"size", ty(T::F)}, {"speed", ty(T::F)}}, SC,
                 [](Parser_ &p, std::vector<Value> &a) { auto c = std::make_shared<SurfaceColor>(); c->kind = EU_COL_PERLIN; c->seed = p.opts.random_seed; c->v[0] = a[0].num; c->v[1] = a[1].num; return wrap(c); });
        }
        addn({"reflection_ratio_uniform_#"}, {{"ratio", ty(T::F)}}, RR, [](Parser_ &, std::vector<Value> &a) { auto r = std::make_shared<ReflectionRatio>(); r->kind = EU_RATIO_UNIFORM; r->p0 = a[0].num; return wrap(r); });
        addn({"reflection_ratio_fresnel_#"}, {{"refractive_index_inside", ty(T::F)}, {"refractive_index_outside", ty(T::F)}}, RR,
             [](Parser_ &, std::vector<Value> &a) { auto r = std::make_shared<ReflectionRatio>(); r->kind = EU_RATIO_FRESNEL; r->p0 = a[0].num; r->p1 = a[1].num; return wrap(r); });
        addn({"reflection_direction_specular_#"}, {}, RD, [](Parser_ &, std::vector<Value> &) { return wrap(std::make_shared<ReflectionDirection>()); });
        addn({"threshold_direction_snell_#"}, {{"refractive_index", ty(T::F)}}, TD, [](Parser_ &, std::vector<Value> &a) { auto t = std::make_shared<ThresholdDirection>(); t->kind = EU_THR_SNELL; t->p0 = a[0].num; return wrap(t); });
        addn({"threshold_direction_identity_#"}, {}, TD, [](Parser_ &, std::vector<Value> &) { auto t = std::make_shared<ThresholdDirection>(); t->kind = EU_THR_IDENTITY; return wrap(t); });
        addn({"surface_color_uniform_#"}, {{"color", ty(T::Rgba)}}, SC, [d](Parser_ &, std::vector<Value> &a) { auto c = std::make_shared<SurfaceColor>(); c->kind = EU_COL_UNIFORM; c->dim = d; for (int i = 0; i < 4; i++) c->c0[i] = a[0].vec[i]; return wrap(c); });
        addn({"surface_color_texture_#"}, {{"mapped_texture", ty(T::MappedTexture, d)}}, SC, [d](Parser_ &, std::vector<Value> &a) { auto c = std::make_shared<SurfaceColor>(); c->kind = EU_COL_TEXTURE; c->dim = d; c->mapped = as<MappedTexture>(a[0]); return wrap(c); });
        /* environments + cameras, scene.rs:1339-1408 */
        addn({"Universe#", "Universe#::new"}, {{"camera", ty(T::Camera, d)}, {"entities", ty(T::Entity, d, true)}, {"background", ty(T::MappedTexture, d)}}, ty(T::Environment),
             [d](Parser_ &, std::vector<Value> &a) {
                 auto u = std::make_shared<Universe>(); u->dim = d;
                 u->camera = *as<eu_camera>(a[0]);
                 for (auto &v : a[1].list) u->entities.push_back(as<Entity>(v));
                 u->background = as<MappedTexture>(a[2]);
                 return wrap(u);
             });
        auto camk = [d](uint32_t kind, bool with_loc) {
            return [d, kind, with_loc](Parser_ &, std::vector<Value> &a) {
                auto c = std::make_shared<eu_camera>(default_camera(d, with_loc ? vp(a[0]) : nullptr));
                c->kind = kind;
                return wrap(c);
            };
        };
        if (d == 3) {      /* scene.rs:1353-1391 */
            addn({"PitchYawCamera3", "PitchYawCamera3::new"}, {}, ty(T::Camera, 3), camk(EU_CAMERA_PITCH_YAW_3, false));
            addn({"PitchYawCamera3::new_with_location"}, {{"location", P}}, ty(T::Camera, 3), camk(EU_CAMERA_PITCH_YAW_3, true));
            addn({"FreeCamera3", "FreeCamera3::new"}, {}, ty(T::Camera, 3), camk(EU_CAMERA_FREE_3, false));
            addn({"FreeCamera3::new_with_location"}, {{"location", P}}, ty(T::Camera, 3), camk(EU_CAMERA_FREE_3, true));
        } else {
            addn({"FreeCamera4", "FreeCamera4::new"}, {}, ty(T::Camera, 4), camk(EU_CAMERA_FREE_4, false));
            addn({"FreeCamera4::new_with_location"}, {{"location", P}}, ty(T::Camera, 4), camk(EU_CAMERA_FREE_4, true));
        }
    }
}
}  // namespace

Parser Parser::make_default(const eu_load_opts *opts) {
    Parser p;
    if (opts) p.opts = *opts;
    return p;
}

std::shared_ptr<Universe> Parser::parse(const char *json, size_t len) {   /* scene.rs:1466-1478 */
    Json doc = Json::parse(json, len);
    Parser_ p;
    p.opts = opts;
    p.build_registry();
    Value v = p.construct(ty(T::Environment), doc);
    textures_substituted = p.substituted;
    return as<Universe>(v);
}

/* ------------------------------------------------------------------ flatten */
namespace {
struct Flattener {
    int D;
    std::vector<EuShapeOp> ops;
    std::vector<real> params;
    std::vector<EuFlatEntity> entities;
    std::vector<EuFlatMaterial> materials;
    std::vector<uint64_t> transforms;     /* 8 words each */
    std::vector<uint64_t> code;
    std::vector<EuFlatSurface> surfaces;
    std::vector<EuFlatColorOp> color_ops;
    std::vector<EuFlatMapped> mapped;
    std::vector<std::shared_ptr<Texture>> textures;
    std::vector<std::array<uint8_t, 512>> perlin;
    std::map<const void *, uint32_t> mat_ids, surf_ids, mapped_ids;
    uint32_t hit_cap = 0, hit_cap_strict = 0, list_depth = 0, color_depth = 0, rpn_depth = 0, n_leaves = 0;
    /* Hit-stack entries, counted twice.  `strict` is the worst case whatever the arithmetic does: a chain of n half-spaces may in
     * principle emit n hits.  `soft` is what the wavefront kernels reserve: an Intersection chain is a convex solid and a line meets
     * its boundary at most twice, so its stream has at most two elements unless rounding noise among near-tied plane hits lets
     * a third through.  A kernel that meets such a ray (its stack fills up: push_chain, csg_merge) raises EuDevCounters::overflow and
     * the frame is traced again by the stack kernel, whose hit stack has the strict size: exact either way, and 4d_frame's stack
     * (strict 80 entries, soft 20) moves from scratch memory into LDS. */
    struct HitUse { uint32_t soft, strict; };

    static bool is_flat_leaf(const Shape &s) { return s.kind == Shape::HalfSpace || s.kind == Shape::Hyperplane; }

    /* left-fold chain of half-space / hyperplane leaves with one Union or Intersection operation
     * (ComposableShape::of, shape.rs:523-545): collects the leaves in fold order */
    static bool collect_chain(const Shape &s, SetOperation op, std::vector<const Shape *> &leaves) {
        if (is_flat_leaf(s)) { leaves.push_back(&s); return true; }
        if (s.kind != Shape::ComposableShape || s.operation != op || !is_flat_leaf(*s.sb)) return false;
        if (!collect_chain(*s.sa, op, leaves)) return false;
        leaves.push_back(s.sb.get());
        return true;
    }

    /* EU_SH_CHAIN_BOX: 2*D half-spaces, leaf k's normal is +-e_(k/2) exactly, its constant finite (0: not a box, 1: all constants
     * non-zero, 2: some constant is +-0 -> EU_SH_CHAIN_BOX0) */
    int is_axis_box(const std::vector<const Shape *> &chain) const {
        if (no_box_chains() || (int)chain.size() != 2 * D) return 0;
        bool zero_c = false;
        for (int k = 0; k < 2 * D; k++) {
            const Shape &l = *chain[k];
            if (l.kind != Shape::HalfSpace) return 0;
            for (int i = 0; i < D; i++) {
                if (i == k / 2) { if (!(l.a[i] == R(1.0) || l.a[i] == -R(1.0))) return 0; }
                else if (!(l.a[i] == R(0.0))) return 0;
            }
            if (!std::isfinite(l.r)) return 0;
            if (l.r == R(0.0)) zero_c = true;
            if (!(l.signum == R(1.0) || l.signum == -R(1.0))) return 0;      /* (the device compares sign bits, chain_matrices_box) */
        }
        /* (zero constants: 3-D only -- the 4-D kernels sit at their register limit and pay for the second variant with spills) */
        return zero_c ? ((D != 3 || diag_env("EU_NO_BOX0_CHAINS")) ? 0 : 2) : 1;
    }
    /* A/B diagnostics: only a -DEU_DIAGNOSTICS build reads the environment (the shipped library reads none but the cache directory's) */
    static bool diag_env(const char *name) {
#ifdef EU_DIAGNOSTICS
        return getenv(name) != nullptr;
#else
        (void)name;
        return false;
#endif
    }
    static bool no_box_chains() { static const bool v = diag_env("EU_NO_BOX_CHAINS"); return v; }

    void push_halfspace_params(const Shape &s) {
        for (int i = 0; i < D; i++) params.push_back(s.a[i]);
        params.push_back(s.r);
        if (s.kind == Shape::HalfSpace) {
            params.push_back(s.signum);
            real k = -s.signum;                                   /* shape.rs:860 normal *= -signum */
            for (int i = 0; i < D; i++) params.push_back(s.a[i] * k);
        } else {                                                    /* Hyperplane: never "inside", normal as given */
            params.push_back(std::nan(""));
            for (int i = 0; i < D; i++) params.push_back(s.a[i]);
        }
    }

    /* returns (max list length of the node's stream); tracks the simulated hit-stack */
    static bool no_guards() { static const bool v = diag_env("EU_NO_SKIP_OPS"); return v; }
    HitUse emit_shape(const Shape &s, HitUse base_use, uint32_t depth, bool is_root = false, real parent_r = INFINITY) {
        if (s.dim != D) fail(ParserError::CustomError, "shape dimension does not match the universe");
        EuShapeOp op{};
        op.first = (uint16_t)ops.size();
        HitUse len{0, 0};
        auto note = [&](HitUse use) { if (use.soft > hit_cap) hit_cap = use.soft; if (use.strict > hit_cap_strict) hit_cap_strict = use.strict; };
        std::vector<const Shape *> chain;
        if (s.kind == Shape::ComposableShape && (s.operation == SetOperation::Union || s.operation == SetOperation::Intersection) &&
            collect_chain(s, s.operation, chain) && chain.size() >= 2 && chain.size() <= EU_CHAIN_MAX) {
            op.kind = (uint8_t)(s.operation == SetOperation::Union ? EU_SH_CHAIN_UNION : EU_SH_CHAIN_INTERSECTION);
            if (s.operation == SetOperation::Intersection) { const int b = is_axis_box(chain); if (b) op.kind = (uint8_t)(b == 2 ? EU_SH_CHAIN_BOX0 : EU_SH_CHAIN_BOX); }
            op.count = (uint8_t)chain.size();
            op.param = (uint32_t)params.size();
            for (auto *leaf : chain) push_halfspace_params(*leaf);
            {   /* the chain's own bounding sphere (axis-aligned boxes only), right after its leaves: c[D], r2, far2; r2 < 0: none */
                Bound b;
                if (s.operation == SetOperation::Intersection) b = box_bound(chain);
                real cmax = R(0.0);
                for (int i = 0; i < D; i++) cmax = std::max(cmax, fabs(b.c[i]));
                const bool ok = b.ok && b.r > R(0.0) && cmax <= R(1.0e6) * b.r;
                const real rr = b.r * (R(1.0) + EU_BOUND_REL) + EU_BOUND_ABS * std::max(R(1.0), cmax);
                for (int i = 0; i < D; i++) params.push_back(ok ? b.c[i] : R(0.0));
                params.push_back(ok ? rr * rr : -R(1.0));
                params.push_back(ok ? EU_BOUND_FAR2 * rr * rr : R(0.0));
            }
            n_leaves += (uint32_t)chain.size();
            const uint32_t cn = (uint32_t)chain.size();
            if (s.operation == SetOperation::Intersection) {  /* the list only (its t are picked out of registers: chain_pick_t) */
                len = HitUse{cn < 2u ? cn : 2u, cn};
                note(HitUse{base_use.soft + len.soft, base_use.strict + len.strict});
            } else {                                          /* the list + slots for the t_k (picked by run-time index) */
                len = HitUse{cn, cn};
                note(HitUse{base_use.soft + 2 * cn, base_use.strict + 2 * cn});
            }
            if (depth + 1 > list_depth) list_depth = depth + 1;
        } else if (s.kind == Shape::ComposableShape) {
            /* a bounded subtree whose sphere is clearly smaller than what encloses it gets a guard op in front (EU_SH_SKIP) */
            const Bound sb_ = shape_bound(s);
            const bool has_b = sb_.ok && sb_.r > R(0.0) && std::isfinite(sb_.r);
            size_t skip_at = (size_t)-1;
            if (!is_root && has_b && !no_guards() && sb_.r < R(0.7) * parent_r) {
                const uint32_t id = register_bound(sb_);
                if (id != 0xffffffffu) {
                    skip_at = ops.size();
                    EuShapeOp sk{};
                    sk.kind = (uint8_t)EU_SH_SKIP; sk.param = id;
                    ops.push_back(sk);
                }
            }
            const real child_r = has_b ? std::min(parent_r, (real)sb_.r) : parent_r;
            const HitUse la = emit_shape(*s.sa, base_use, depth, false, child_r);
            const HitUse lb = emit_shape(*s.sb, HitUse{base_use.soft + la.soft, base_use.strict + la.strict}, depth + 1, false, child_r);
            if (skip_at != (size_t)-1) {
                if (ops.size() >= 65535) fail(ParserError::CustomError, "too many shape nodes");
                ops[skip_at].first = (uint16_t)ops.size();        /* index the subtree's root op is about to get */
                op.count = 1;                                      /* "my first op is my guard" (eval_shape: child a starts behind it) */
            }
            /* a Complement may hand out `a` once more without consuming it (shape.rs:390-392): one element more than it
             * consumed.  At an entity's root only element 0 is ever looked at, so the extra slot is not reserved there. */
            const uint32_t extra = s.operation == SetOperation::Complement ? 1u : 0u;
            len = HitUse{la.soft + lb.soft + extra, la.strict + lb.strict + extra};
            /* inputs + merge output (csg_merge).  At an entity's root the merge stops at its first element: one slot.  A right operand
             * of at most two hits is read into registers and the output is written over it. */
            auto merge_use = [&](uint32_t a, uint32_t b) {
                const uint32_t out = is_root ? 1u : a + b + extra;
                return a + (b <= 2u ? std::max(b, out) : b + out);
            };
            note(HitUse{base_use.soft + merge_use(la.soft, lb.soft), base_use.strict + merge_use(la.strict, lb.strict)});
            if (depth + 2 > list_depth) list_depth = depth + 2;
            op.kind = (uint8_t)(EU_SH_UNION + (int)s.operation);
            op.param = 0;
        } else {
            n_leaves++;
            op.param = (uint32_t)params.size();
            switch (s.kind) {
            case Shape::VoidShape: op.kind = EU_SH_VOID; len = HitUse{0, 0}; break;
            case Shape::Sphere:
                op.kind = EU_SH_SPHERE; len = HitUse{2, 2};
                for (int i = 0; i < D; i++) params.push_back(s.a[i]);
                params.push_back(s.r); params.push_back(s.r * s.r);
                break;
            case Shape::Hyperplane:
                op.kind = EU_SH_PLANE; len = HitUse{1, 1};
                for (int i = 0; i < D; i++) params.push_back(s.a[i]);
                params.push_back(s.r);
                break;
            case Shape::HalfSpace:
                op.kind = EU_SH_HALFSPACE; len = HitUse{1, 1};
                push_halfspace_params(s);
                break;
            default:
                op.kind = EU_SH_CYLINDER; len = HitUse{2, 2};
                for (int i = 0; i < D; i++) params.push_back(s.a[i]);
                for (int i = 0; i < D; i++) params.push_back(s.b[i]);
                params.push_back(s.r); params.push_back(s.r * s.r);
                break;
            }
            note(HitUse{base_use.soft + len.soft, base_use.strict + len.strict});
            if (depth + 1 > list_depth) list_depth = depth + 1;
        }
        if (ops.size() >= 65535) fail(ParserError::CustomError, "too many shape nodes");
        ops.push_back(op);
        return len;
    }

    /* ---- conservative bounding spheres (exact culling, DESIGN.md "Culling") ---- */
    struct Bound { bool ok = false; real c[MAXD] = {0, 0, 0, 0}; real r = R(0.0); };
    std::vector<real> bounds;

    static Bound enclose(int D, const Bound &a, const Bound &b) {
        Bound o;
        if (!a.ok || !b.ok) return o;
        real dist2 = R(0.0);
        for (int i = 0; i < D; i++) dist2 += (a.c[i] - b.c[i]) * (a.c[i] - b.c[i]);
        const real dist = sqrt(dist2);
        if (dist + b.r <= a.r) return a;
        if (dist + a.r <= b.r) return b;
        o.ok = true;
        o.r = (dist + a.r + b.r) / R(2.0);
        for (int i = 0; i < D; i++) o.c[i] = a.c[i] + (b.c[i] - a.c[i]) * ((o.r - a.r) / dist);
        return o;
    }

    /* an Intersection chain of axis-aligned half-spaces that bounds every axis on both sides (cuboid, hypercuboid) */
    Bound box_bound(const std::vector<const Shape *> &leaves) const {
        Bound o;
        real lo[MAXD], hi[MAXD]; bool has_lo[MAXD] = {false, false, false, false}, has_hi[MAXD] = {false, false, false, false};
        for (auto *s : leaves) {
            if (s->kind != Shape::HalfSpace || !(s->signum == R(1.0) || s->signum == -R(1.0))) return o;
            int axis = -1;
            for (int i = 0; i < D; i++) if (s->a[i] != R(0.0)) { if (axis >= 0) return o; axis = i; }
            if (axis < 0) return o;
            const real sn = s->a[axis], edge = -s->r / sn;          /* inside <=> sign(sn*x + c) == signum */
            if (!std::isfinite(edge)) return o;
            if (s->signum * sn > R(0.0)) { if (!has_lo[axis] || edge > lo[axis]) lo[axis] = edge; has_lo[axis] = true; }
            else { if (!has_hi[axis] || edge < hi[axis]) hi[axis] = edge; has_hi[axis] = true; }
        }
        real r2 = R(0.0);
        for (int i = 0; i < D; i++) {
            if (!has_lo[i] || !has_hi[i] || !(hi[i] >= lo[i])) return o;
            o.c[i] = (lo[i] + hi[i]) / R(2.0);
            r2 += ((hi[i] - lo[i]) / R(2.0)) * ((hi[i] - lo[i]) / R(2.0));
        }
        o.r = sqrt(r2); o.ok = true;
        return o;
    }

    /* Intersection tree of ONE infinite cylinder with half-spaces whose normals are parallel to its axis and that cut it on both
     * sides (Cylinder::new_with_height, shape.rs:906-927): centre on the axis between the caps, radius^2 = r^2 + (half length)^2 */
    static void intersection_leaves(const Shape &s, std::vector<const Shape *> &out) {
        if (s.kind == Shape::ComposableShape && s.operation == SetOperation::Intersection) { intersection_leaves(*s.sa, out); intersection_leaves(*s.sb, out); }
        else out.push_back(&s);
    }
    Bound capped_cylinder_bound(const Shape &s) const {
        Bound o;
        std::vector<const Shape *> leaves;
        intersection_leaves(s, leaves);
        const Shape *cyl = nullptr;
        for (auto *l : leaves) { if (l->kind == Shape::Cylinder) { if (cyl) return o; cyl = l; } else if (l->kind != Shape::HalfSpace) return o; }
        if (!cyl || !(cyl->r > R(0.0)) || !std::isfinite(cyl->r)) return o;
        real u[MAXD], un = R(0.0);
        for (int i = 0; i < D; i++) un += cyl->b[i] * cyl->b[i];
        un = sqrt(un);
        if (!(un > R(0.0)) || !std::isfinite(un)) return o;
        for (int i = 0; i < D; i++) u[i] = cyl->b[i] / un;
        bool has_lo = false, has_hi = false; real lo = R(0.0), hi = R(0.0);
        for (auto *l : leaves) {
            if (l == cyl) continue;
            if (!(l->signum == R(1.0) || l->signum == -R(1.0))) return o;
            real k1 = R(0.0), k0 = l->r, nn = R(0.0);
            for (int i = 0; i < D; i++) { k1 += l->a[i] * u[i]; k0 += l->a[i] * cyl->a[i]; nn += l->a[i] * l->a[i]; }
            real perp2 = nn - k1 * k1;                          /* the normal's part across the axis */
            if (!(nn > R(0.0)) || !(perp2 <= R(1.0e-18) * nn) || k1 == R(0.0)) continue;      /* not a cap: ignored (it can only cut more away) */
            const real edge = -k0 / k1;                          /* inside <=> sign(k0 + k1 * tau) == signum */
            if (!std::isfinite(edge)) continue;
            if (l->signum * k1 > R(0.0)) { if (!has_lo || edge > lo) lo = edge; has_lo = true; }
            else { if (!has_hi || edge < hi) hi = edge; has_hi = true; }
        }
        if (!has_lo || !has_hi || !(hi > lo)) return o;
        const real mid = (lo + hi) / R(2.0), half = (hi - lo) / R(2.0);
        for (int i = 0; i < D; i++) o.c[i] = cyl->a[i] + u[i] * mid;
        o.r = sqrt(cyl->r * cyl->r + half * half);
        o.ok = std::isfinite(o.r);
        return o;
    }

    Bound shape_bound(const Shape &s) const {
        Bound o;
        switch (s.kind) {
        case Shape::Sphere:
            if (!(s.r > R(0.0)) || !std::isfinite(s.r)) return o;
            o.ok = true; o.r = s.r;
            for (int i = 0; i < D; i++) o.c[i] = s.a[i];
            return o;
        case Shape::ComposableShape: {
            if (s.operation == SetOperation::Intersection) {
                std::vector<const Shape *> chain;
                if (collect_chain(s, SetOperation::Intersection, chain)) { Bound b = box_bound(chain); if (b.ok) return b; }
                { Bound cb = capped_cylinder_bound(s); if (cb.ok) return cb; }
                Bound a = shape_bound(*s.sa), b = shape_bound(*s.sb);
                if (a.ok && b.ok) return a.r <= b.r ? a : b;
                return a.ok ? a : b;
            }
            if (s.operation == SetOperation::Complement) return shape_bound(*s.sa);
            return enclose(D, shape_bound(*s.sa), shape_bound(*s.sb));       /* Union, SymmetricDifference */
        }
        default: return o;     /* half-spaces, planes, infinite cylinders, VoidShape */
        }
    }

    uint32_t entity_bound(const Shape &s) { return register_bound(shape_bound(s)); }
    uint32_t register_bound(const Bound &b) {
        if (!b.ok || !(b.r > R(0.0))) return 0xffffffffu;
        real cmax = R(0.0);
        for (int i = 0; i < D; i++) { if (!std::isfinite(b.c[i])) return 0xffffffffu; cmax = std::max(cmax, fabs(b.c[i])); }
        if (cmax > R(1.0e6) * b.r) return 0xffffffffu;          /* the margin below must dominate rounding of |o - c|^2 */
        const real rr = b.r * (R(1.0) + EU_BOUND_REL) + EU_BOUND_ABS * std::max(R(1.0), cmax);
        uint32_t id = (uint32_t)(bounds.size() / (size_t)(D + 2));
        for (int i = 0; i < D; i++) bounds.push_back(b.c[i]);
        bounds.push_back(rr * rr);
        bounds.push_back(EU_BOUND_FAR2 * rr * rr);
        return id;
    }

    uint64_t emit_expr(const Expr &e, const std::string &legend) {
        uint32_t off = (uint32_t)code.size();
        for (auto &t : e.rpn) {
            uint32_t arg = t.arg;
            if (t.op == EU_RPN_VAR) {
                /* material.rs:76-89: one legend character per component; an unknown variable would
                 * panic at the first evaluation ("Could not evaluate the expression.") */
                int idx = -1;
                if (t.var.size() == 1) for (int i = 0; i < D && i < (int)legend.size(); i++) if (legend[(size_t)i] == t.var[0]) idx = i;
                if (idx < 0) fail(ParserError::CustomError, "Could not evaluate the expression. (unknown variable `" + t.var + "`)");
                arg = (uint32_t)idx;
            }
            code.push_back((uint64_t)t.op | ((uint64_t)arg << 32));
            if (t.op == EU_RPN_CONST) { uint64_t w; memcpy(&w, &t.k, 8); code.push_back(w); }
        }
        uint32_t len = (uint32_t)code.size() - off;
        uint32_t sd = (uint32_t)e.stack_depth();
        if (sd > rpn_depth) rpn_depth = sd;
        return (uint64_t)off | ((uint64_t)len << 32);
    }

    uint32_t material_id(const MaterialPtr &m) {
        auto it = mat_ids.find(m.get());
        if (it != mat_ids.end()) return it->second;
        if (m->dim != D) fail(ParserError::CustomError, "material dimension does not match the universe");
        EuFlatMaterial fm{};
        if (m->kind == Material::LinearSpace) {
            if ((int)m->legend.size() < D) fail(ParserError::CustomError, "The legend is too short! Make sure it is sufficient for " + std::to_string(D) + " dimensions.");
            fm.first_transform = (uint32_t)(transforms.size() / 8);
            for (auto &tr : m->transformations) {
                if ((int)tr->expressions.size() != D)
                    fail(ParserError::CustomError, "The number of functions must be equal to the number of dimensions (" + std::to_string(D) + ")!");
                uint64_t rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < D; i++) { rec[i] = emit_expr(tr->expressions[(size_t)i].expression, m->legend); rec[4 + i] = emit_expr(tr->expressions[(size_t)i].inverse_expression, m->legend); }
                transforms.insert(transforms.end(), rec, rec + 8);
            }
            fm.kind = EU_MAT_LINEAR | ((uint32_t)m->transformations.size() << 8);
        } else fm.kind = EU_MAT_VACUUM;
        uint32_t id = (uint32_t)materials.size();
        materials.push_back(fm);
        mat_ids[m.get()] = id;
        return id;
    }

    uint32_t mapped_id(const std::shared_ptr<MappedTexture> &m) {
        auto it = mapped_ids.find(m.get());
        if (it != mapped_ids.end()) return it->second;
        EuFlatMapped fm{};
        fm.tex_kind = m->texture->kind; fm.uv_kind = 0;
        fm.w = m->texture->w; fm.h = m->texture->h; fm.texels = 0;
        for (int i = 0; i < 3; i++) fm.center[i] = m->uvfn->center[i];
        fm.wd = (real)fm.w; fm.hd = (real)fm.h;
        uint32_t id = (uint32_t)mapped.size();
        mapped.push_back(fm); textures.push_back(m->texture);
        mapped_ids[m.get()] = id;
        return id;
    }

    static void perlin_perm(uint32_t seed, uint8_t *perm512) {
        uint8_t p[256];
        for (int i = 0; i < 256; i++) p[i] = (uint8_t)i;
        uint32_t st = seed + 0x9E3779B9u;
        if (st == 0) st = 1;
        for (int i = 255; i >= 1; i--) {
            st ^= st << 13; st ^= st >> 17; st ^= st << 5;
            uint32_t j = st % (uint32_t)(i + 1);
            uint8_t tmp = p[i]; p[i] = p[j]; p[j] = tmp;
        }
        for (int i = 0; i < 512; i++) perm512[i] = p[i & 255];
    }

    /* post-order colour program; returns stack depth needed */
    uint32_t emit_color(const SurfaceColor &c) {
        EuFlatColorOp op{};
        op.kind = c.kind;
        uint32_t depth = 1;
        switch (c.kind) {
        case EU_COL_BLEND: {
            uint32_t da = emit_color(*c.source);
            uint32_t db = emit_color(*c.destination);
            depth = da > db + 1 ? da : db + 1;
            op.fn = c.blend->fn; op.v[0] = c.blend->ratio;
            break;
        }
        case EU_COL_UNIFORM: for (int i = 0; i < 4; i++) op.c0[i] = c.c0[i]; break;      /* (the flat records hold f64 whatever F is: element-wise, widening) */
        case EU_COL_ILLUM_GLOBAL: for (int i = 0; i < 4; i++) { op.c0[i] = c.c0[i]; op.c1[i] = c.c1[i]; } break;
        case EU_COL_ILLUM_DIR:
            for (int i = 0; i < 4; i++) { op.c0[i] = c.c0[i]; op.c1[i] = c.c1[i]; op.v[i] = c.v[i]; }
            for (int i = 0; i < D; i++) op.v[i] = -c.v[i];      /* surface.rs:403 uses -light_direction */
            break;
        case EU_COL_PERLIN: {
            if (D != 3) fail(ParserError::CustomError, "surface_color_perlin_hue is 3-D only");
            std::array<uint8_t, 512> perm;
            perlin_perm(c.seed, perm.data());
            op.aux = (uint32_t)perlin.size();
            perlin.push_back(perm);
            op.v[0] = c.v[0]; op.v[1] = c.v[1];
            break;
        }
        default: op.aux = mapped_id(c.mapped); break;
        }
        color_ops.push_back(op);
        return depth;
    }

    uint32_t surface_id(const std::shared_ptr<ComposableSurface> &s) {
        auto it = surf_ids.find(s.get());
        if (it != surf_ids.end()) return it->second;
        EuFlatSurface fs{};
        fs.ratio_kind = s->reflection_ratio->kind; fs.ratio_p0 = s->reflection_ratio->p0; fs.ratio_p1 = s->reflection_ratio->p1;
        fs.thr_kind = s->threshold_direction->kind; fs.thr_p0 = s->threshold_direction->p0;
        fs.thr_p0_inv = R(1.0) / s->threshold_direction->p0;       /* surface.rs:278 */
        fs.color_first = (uint32_t)color_ops.size();
        uint32_t depth = emit_color(*s->surface_color);
        if (depth > color_depth) color_depth = depth;
        fs.color_root = (uint32_t)color_ops.size() - 1;
        uint32_t id = (uint32_t)surfaces.size();
        surfaces.push_back(fs);
        surf_ids[s.get()] = id;
        return id;
    }
};
}  // namespace

FlatScene flatten(const Universe &u) {
    Flattener f;
    f.D = u.dim;
    if (u.dim != 3 && u.dim != 4) fail(ParserError::CustomError, "only 3-D and 4-D universes exist");
    if (!u.background) fail(ParserError::CustomError, "the universe has no background");
    for (auto &e : u.entities) {
        EuFlatEntity fe{};
        uint32_t before_cap = f.hit_cap;
        f.hit_cap = 0;
        fe.shape_first = (uint16_t)f.ops.size();
        {
            const auto eb = f.shape_bound(*e->shape);
            f.emit_shape(*e->shape, Flattener::HitUse{0, 0}, 0, true, eb.ok && eb.r > R(0.0) ? (real)eb.r : (real)INFINITY);
        }
        fe.shape_root = (uint16_t)(f.ops.size() - 1);
        fe.max_hits = f.hit_cap;
        fe.bound = e->surface ? f.entity_bound(*e->shape) : 0xffffffffu;
        if (before_cap > f.hit_cap) f.hit_cap = before_cap;
        fe.material = (uint16_t)f.material_id(e->material);
        fe.surface = e->surface ? (int16_t)f.surface_id(e->surface) : (int16_t)-1;
        f.entities.push_back(fe);
    }
    /* the flat records index with 16 bits (EuFlatEntity, hit codes, EuShapeOp::first): refuse rather than wrap around */
    if (f.ops.size() > 0xfff0u || f.entities.size() > 0xfff0u || f.materials.size() > 0x7ff0u || f.surfaces.size() > 0x7ff0u)
        fail(ParserError::CustomError, "the scene exceeds the flat format's 16-bit indices (shape operations, entities, materials or surfaces)");
    uint32_t bg = f.mapped_id(u.background);
    if (u.background->dim != u.dim) fail(ParserError::CustomError, "background dimension mismatch");

    FlatScene out;
    auto &w = out.words;
    w.resize(EU_FLAT_HEADER_WORDS, 0);
    EuFlatHeader h{};
    h.magic = EU_FLAT_MAGIC; h.version = EU_FLAT_VERSION; h.dim = (uint32_t)u.dim;
    auto append = [&](const void *data, size_t bytes) -> uint32_t {
        uint32_t off = (uint32_t)w.size();
        size_t nw = (bytes + 7) / 8;
        w.resize(w.size() + nw, 0);
        if (bytes) memcpy(&w[off], data, bytes);
        return off;
    };
    h.n_ops = (uint32_t)f.ops.size(); h.off_ops = append(f.ops.data(), f.ops.size() * sizeof(EuShapeOp));
    h.n_params = (uint32_t)f.params.size(); h.off_params = append(f.params.data(), f.params.size() * sizeof(f.params[0]));
    h.n_entities = (uint32_t)f.entities.size(); h.off_entities = append(f.entities.data(), f.entities.size() * sizeof(EuFlatEntity));
    h.n_materials = (uint32_t)f.materials.size(); h.off_materials = append(f.materials.data(), f.materials.size() * sizeof(EuFlatMaterial));
    h.n_transforms = (uint32_t)(f.transforms.size() / 8); h.off_transforms = append(f.transforms.data(), f.transforms.size() * 8);
    h.n_code = (uint32_t)f.code.size(); h.off_code = append(f.code.data(), f.code.size() * 8);
    h.n_surfaces = (uint32_t)f.surfaces.size(); h.off_surfaces = append(f.surfaces.data(), f.surfaces.size() * sizeof(EuFlatSurface));
    h.n_color_ops = (uint32_t)f.color_ops.size(); h.off_color_ops = append(f.color_ops.data(), f.color_ops.size() * sizeof(EuFlatColorOp));
    h.n_mapped = (uint32_t)f.mapped.size(); h.off_mapped = append(f.mapped.data(), f.mapped.size() * sizeof(EuFlatMapped));
    h.n_perlin = (uint32_t)f.perlin.size(); h.off_perlin = append(f.perlin.data(), f.perlin.size() * 512);
    h.n_bounds = (uint32_t)(f.bounds.size() / (size_t)(u.dim + 2)); h.off_bounds = append(f.bounds.data(), f.bounds.size() * sizeof(f.bounds[0]));
    h.background = bg; h.hit_cap = f.hit_cap | (f.hit_cap_strict << 16); h.list_depth = f.list_depth; h.color_depth = f.color_depth; h.rpn_depth = f.rpn_depth;
    /* flags bit 1: some shape program holds guard ops (EU_SH_SKIP) */
    for (auto &o : f.ops) if (o.kind == EU_SH_SKIP) { h.flags |= 2u; break; }
    /* flags bit 0: some surface can spawn BOTH a transmission and a reflection ray (ratio strictly between 0 and 1
     * possible): the recursion tree branches and a frame holds several times more rays than pixels */
    for (auto &fs : f.surfaces)
        if (fs.ratio_kind == EU_RATIO_FRESNEL || (fs.ratio_p0 > R(0.0) && fs.ratio_p0 < R(1.0))) h.flags |= 1u;
    /* flags bit 2: NO surface ever asks for a reflection (every ratio provider is the uniform one with a ratio that is not > 0; surface.rs:
     * 119-139, 200-211): then no node of the recursion tree waits for two children and the renderer launches no resolve passes */
    {
        bool reflects = false;
        for (auto &fs : f.surfaces) if (fs.ratio_kind != EU_RATIO_UNIFORM || !(fs.ratio_p0 <= R(0.0))) reflects = true;
        if (!reflects) h.flags |= 4u;
    }
    h.n_words = (uint32_t)w.size();
    memcpy(w.data(), &h, sizeof h);
    out.textures = f.textures;
    out.info.dim = u.dim;
    out.info.n_entities = h.n_entities; out.info.n_shape_ops = h.n_ops; out.info.n_leaves = f.n_leaves;
    out.info.n_materials = h.n_materials; out.info.n_surfaces = h.n_surfaces; out.info.n_color_ops = h.n_color_ops;
    out.info.n_textures = h.n_mapped; out.info.hit_cap = f.hit_cap_strict; out.info.list_depth = h.list_depth;
    out.info.flat_bytes = h.n_words * 8;
    return out;
}

}  // namespace euclider
