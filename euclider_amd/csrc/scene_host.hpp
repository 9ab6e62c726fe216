/*
 * scene_host.hpp -- host-side universe model, JSON scene parser and flattener (C++17).
 *
 * Mirrors the reference's loader surface for the trace path (names, field order, aliases and
 * error taxonomy of /root/reference/src/scene.rs:524-552,564-1478) and the load-time arithmetic
 * of the shape/material constructors (universe/entity/shape.rs:523-545,750-766,828-841,893-927;
 * d3/entity/shape.rs:17-66; d4/entity/shape.rs:18-76).  The objects are descriptions only: all
 * tracing happens in the HIP kernels (trace_kernel.hip); there is no host trace path.
 */
#ifndef EU_SCENE_HOST_HPP
#define EU_SCENE_HOST_HPP

/* every system header the loader's translation units use comes first: eu_real.h (last include below) may redefine `double` */
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/euclider_amd.h"
#include "flat_scene.h"
#include "eu_math.h"
#include "eu_real.h"

namespace euclider {

constexpr int MAXD = 4;

/* scene.rs:524-552 */
struct ParserError {
    enum Kind { NoDeserializer, SyntaxError, MissingType, InvalidConstructor, MissingField, TypeMismatch, CustomError };
    Kind kind;
    std::string description;
    static const char *kind_name(Kind k);
    std::string what() const { return std::string(kind_name(kind)) + ": " + description; }
};

/* ---- minimal JSON document (ordered objects, like the `json` crate's Object) ---- */
struct Json {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    eu_f64 num = R(0.0);        /* JSON numbers are f64 (json 0.11); a field of type F narrows it */
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json *get(const std::string &key) const;
    static Json parse(const char *text, size_t len);   /* throws ParserError{SyntaxError} */
    std::string brief() const;
};

/* ---- universe model ---- */
enum class SetOperation { Union = 0, Intersection = 1, Complement = 2, SymmetricDifference = 3 };

struct Shape {
    enum Kind { VoidShape, Sphere, Hyperplane, HalfSpace, Cylinder, ComposableShape } kind = VoidShape;
    int dim = 3;
    real a[MAXD] = {0, 0, 0, 0};   /* sphere centre | plane normal | cylinder centre */
    real b[MAXD] = {0, 0, 0, 0};   /* cylinder axis (normalised) */
    real r = R(0.0);                  /* radius | plane constant */
    real signum = R(0.0);             /* HalfSpace */
    SetOperation operation = SetOperation::Union;
    std::shared_ptr<Shape> sa, sb;
};
using ShapePtr = std::shared_ptr<Shape>;

/* constructors, named after the reference's */
ShapePtr VoidShape_new(int dim);
ShapePtr Sphere_new(int dim, const real *center, real radius);
ShapePtr Hyperplane_new(int dim, const real *normal, real constant);
ShapePtr Hyperplane_new_with_point(int dim, const real *normal, const real *point);
ShapePtr Hyperplane_new_with_vectors(const real *a, const real *b, const real *point);   /* 3-D */
ShapePtr HalfSpace_new(const ShapePtr &plane, real sign);
ShapePtr HalfSpace_new_with_point(const ShapePtr &plane, const real *point);
ShapePtr HalfSpace_cuboid(const real *center, const real *abc);          /* d3::cuboid */
ShapePtr HalfSpace_hypercuboid(const real *center, const real *abcd);    /* d4::hypercuboid */
ShapePtr Cylinder_new(int dim, const real *center, const real *direction, real radius);
ShapePtr Cylinder_new_with_height(int dim, const real *center, const real *direction, real radius, real height);
ShapePtr ComposableShape_of(const std::vector<ShapePtr> &shapes, SetOperation op);

/* meval-subset expression compiled to RPN words (flat_scene.h EuRpn) */
struct Expr {
    std::string source;
    struct Tok { uint32_t op, arg; eu_f64 k; std::string var; };      /* constants of an expression are f64 (meval) */
    std::vector<Tok> rpn;
    static Expr from_str(const std::string &s);     /* throws ParserError{CustomError} */
    int stack_depth() const;
};
struct ComponentTransformationExpr { Expr expression, inverse_expression; };
struct ComponentTransformation { std::vector<ComponentTransformationExpr> expressions; };

struct Material {
    enum Kind { Vacuum, LinearSpace } kind = Vacuum;
    int dim = 3;
    std::string legend;
    std::vector<std::shared_ptr<ComponentTransformation>> transformations;
};
using MaterialPtr = std::shared_ptr<Material>;

struct Texture { uint32_t kind = 0, w = 0, h = 0; std::shared_ptr<std::vector<uint8_t>> rgba; std::string path; };
struct UVFn { int dim = 3; real center[3] = {0, 0, 0}; };   /* uv_sphere_3, optionally wrapped by uv_derank_4 */
struct MappedTexture { int dim = 3; std::shared_ptr<UVFn> uvfn; std::shared_ptr<Texture> texture; };

struct BlendFunction { uint32_t fn = 0; real ratio = R(0.0); };
struct SurfaceColor {
    uint32_t kind = 0;                 /* EuColorKind */
    int dim = 3;
    real c0[4] = {0, 0, 0, 0}, c1[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
    std::shared_ptr<SurfaceColor> source, destination;
    std::shared_ptr<BlendFunction> blend;
    uint32_t seed = 0;
    std::shared_ptr<MappedTexture> mapped;
};
struct ReflectionRatio { uint32_t kind = 0; real p0 = 0, p1 = 0; };
struct ReflectionDirection {};
struct ThresholdDirection { uint32_t kind = 0; real p0 = 0; };
struct ComposableSurface {
    std::shared_ptr<ReflectionRatio> reflection_ratio;
    std::shared_ptr<ReflectionDirection> reflection_direction;
    std::shared_ptr<ThresholdDirection> threshold_direction;
    std::shared_ptr<SurfaceColor> surface_color;
};
struct Entity {
    ShapePtr shape;
    MaterialPtr material;
    std::shared_ptr<ComposableSurface> surface;   /* null: Void / new_without_surface */
};

struct Universe {
    int dim = 3;
    eu_camera camera{};
    std::vector<std::shared_ptr<Entity>> entities;
    std::shared_ptr<MappedTexture> background;
};

eu_camera default_camera(int dim, const real *location_or_null);
void rgba_from_hsva(real hue, real saturation, real value, real alpha, real *out);
void procedural_uv_grid(uint32_t w, uint32_t h, std::vector<uint8_t> &rgba);

/* ---- the parser (scene.rs:554-1478) ---- */
struct Parser {
    eu_load_opts opts{};
    uint32_t textures_substituted = 0;
    static Parser make_default(const eu_load_opts *opts);
    std::shared_ptr<Universe> parse(const char *json, size_t len);   /* throws ParserError */
    struct Impl;
    std::shared_ptr<Impl> impl;
};

/* ---- flattening ---- */
struct FlatScene {
    std::vector<uint64_t> words;                       /* header + tables (flat_scene.h) */
    std::vector<std::shared_ptr<Texture>> textures;    /* one per EuFlatMapped, same order */
    eu_scene_info info{};
    EuFlatHeader &header() { return *reinterpret_cast<EuFlatHeader *>(words.data()); }
    const EuFlatHeader &header() const { return *reinterpret_cast<const EuFlatHeader *>(words.data()); }
};
FlatScene flatten(const Universe &u);   /* throws ParserError{CustomError} on capacity problems */

}  // namespace euclider

struct eu_scene {
    std::shared_ptr<euclider::Universe> universe;
    euclider::FlatScene flat;
};

#endif
