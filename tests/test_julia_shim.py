"""CPU: the Julia binding a maintainer would add (integration/DTOEngine.jl) against include/dto_engine.h.  Julia is not in
the image, so the file cannot run here; what can be checked mechanically is that its plain-C struct layouts, constants and
`@ccall` signatures are those of the header, and that it forwards every MOI method of the reference's evaluator."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = open(os.path.join(ROOT, "integration", "DTOEngine.jl"), encoding="utf-8").read()
H = open(os.path.join(ROOT, "include", "dto_engine.h"), encoding="utf-8").read()
H_NOCOMMENT = re.sub(r"/\*.*?\*/", "", H, flags=re.S)

CTYPE = {"int32_t": "Int32", "int64_t": "Int64", "double": "Float64"}


def c_struct(name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), H_NOCOMMENT, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl)
        base, star, names = m.group(2), m.group(3), m.group(4)
        for nm in names.split(","):
            nm = nm.strip()
            ptr = bool(star) or nm.startswith("*")
            nm = nm.lstrip("* ")
            jt = CTYPE.get(base, {"dto_integrator_desc": "IntegratorDesc", "dto_objective_desc": "ObjectiveDesc",
                                  "dto_constraint_desc": "ConstraintDesc"}.get(base, base))
            out.append((nm, f"Ptr{{{jt}}}" if ptr else jt))
    return out


def jl_struct(name):
    body = re.search(r"^struct %s\n(.*?)^end" % name, JL, flags=re.S | re.M).group(1)
    return [tuple(x.strip() for x in line.split("::")) for line in body.strip().splitlines()]


def test_struct_layouts_match_the_header():
    for jl, c in (("IntegratorDesc", "dto_integrator_desc"), ("ObjectiveDesc", "dto_objective_desc"),
                  ("ConstraintDesc", "dto_constraint_desc"), ("ProblemDesc", "dto_problem_desc"),
                  ("ExternalValues", "dto_external_values"), ("GatherLayout", "dto_gather_layout")):
        assert jl_struct(jl) == c_struct(c), (jl, jl_struct(jl), c_struct(c))


def test_constants_match_the_header():
    defs = dict(re.findall(r"#define (DTO_\w+) (\d+)", H))
    for name, val in re.findall(r"const (DTO_\w+) = Int32\((\d+)\)", JL):
        assert defs[name] == val, name
    consts = dict(re.findall(r"const (DTO_\w+) = Int32\((\d+)\)", JL))
    for need in ("DTO_ABI_VERSION", "DTO_OBJECTIVE_EXTERNAL_GLOBAL", "DTO_CONSTRAINT_EXTERNAL_GLOBAL", "DTO_VECTOR_JACOBIAN",
                 "DTO_VECTOR_HESSIAN", "DTO_VECTOR_GRADIENT", "DTO_VECTOR_CONSTRAINT", "DTO_COMM_ID_BYTES"):
        assert need in consts, need


def _c_prototypes():
    protos = {}
    for ret, name, args in re.findall(r"^(?:const )?(\w+\*?)\s+\*?(dto_\w+)\((.*?)\);", H_NOCOMMENT, flags=re.S | re.M):
        protos[name] = [a.strip() for a in args.replace("\n", " ").split(",") if a.strip() and a.strip() != "void"]
    return protos


def test_every_ccall_names_an_entry_point_with_the_right_argument_count():
    protos = _c_prototypes()
    calls = re.findall(r"@ccall\(?\s*lib\.(dto_\w+)\((.*?)\)::(\w+)", JL, flags=re.S)
    assert len(calls) >= 30
    called = {name for name, _, _ in calls}
    # the device-resident callbacks (MadNLP GPU mode) and the engine's collectives are bound too
    for need in ("dto_eval_objective_dev", "dto_eval_gradient_dev", "dto_eval_constraint_dev", "dto_eval_jacobian_dev",
                 "dto_eval_hessian_dev", "dto_bind_output_dev", "dto_comm_unique_id", "dto_comm_create", "dto_comm_destroy", "dto_get_gather_layout",
                 "dto_gather_jacobian_dev", "dto_gather_hessian_dev", "dto_gather_gradient_dev", "dto_gather_constraint_dev",
                 "dto_allreduce_objective_dev", "dto_interval_costs"):
        assert need in called, need
    for name, args, ret in calls:
        assert name in protos, name
        n_args = len([a for a in re.split(r",(?![^{]*\})", args) if a.strip()])
        assert n_args == len(protos[name]), (name, args, protos[name])
        assert ret in ("Cint", "Cstring", "Cvoid")


def test_every_moi_method_of_the_reference_evaluator_is_forwarded():
    for method in ("initialize", "features_available", "eval_objective", "eval_objective_gradient", "eval_constraint",
                   "jacobian_structure", "eval_constraint_jacobian", "hessian_lagrangian_structure", "eval_hessian_lagrangian",
                   "eval_constraint_jacobian_product", "eval_constraint_jacobian_transpose_product"):
        assert re.search(r"MOI\.%s\(" % method, JL), method
    # generators come from the closure the integrator stores (it has no G field), external terms reach dto_set_external
    assert "B.f.G" in JL and "B.G(" not in JL and "dto_set_external" in JL and "finalizer" in JL
    # the Global* kinds are merged, not refused (global_objectives.jl:35-350, global_constraint.jl:20-160)
    for kind in ("GlobalObjective", "GlobalKnotPointObjective", "NonlinearGlobalConstraint"):
        assert re.search(r"isa %s\b" % kind, JL), kind
    assert "DTO_OBJECTIVE_EXTERNAL_GLOBAL," in JL and "DTO_CONSTRAINT_EXTERNAL_GLOBAL," in JL
    # no elided code: "..." only ever appears as Julia's splat operator, directly behind an expression
    for m in re.finditer(r"\.\.\.", JL):
        assert JL[m.start() - 1] in "])s", JL[max(0, m.start() - 40):m.end() + 5]
