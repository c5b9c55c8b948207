"""bench.py's step, statically: the per-step wait for the frame path and the pose fetch must not depend on developer switches.

(From the middle of round 4 to the middle of round 5 `ctx.sync()` and `po.fetch()` sat under `if step_trace is not None:` -- the default run never
waited for a step's frame path inside the step, and the multi-GPU trajectory gather referred to poses that were never fetched.  bench.py needs a
GPU to run, so the guard is on its syntax tree.)"""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _calls_with_conditions(fn):
    out = []

    def walk(node, conds):
        for child in ast.iter_child_nodes(node):
            if isinstance(child, ast.If):
                test = ast.unparse(child.test)
                for b in child.body:
                    walk_stmt(b, conds + [test])
                for b in child.orelse:
                    walk_stmt(b, conds + ["not (%s)" % test])
            else:
                walk_stmt(child, conds)

    def walk_stmt(node, conds):
        if isinstance(node, ast.If):
            test = ast.unparse(node.test)
            for b in node.body:
                walk_stmt(b, conds + [test])
            for b in node.orelse:
                walk_stmt(b, conds + ["not (%s)" % test])
            return
        for sub in ast.walk(node):
            if isinstance(sub, ast.Call):
                out.append((ast.unparse(sub.func), list(conds)))
        # nested statements with their own bodies (for / with / try) are covered by ast.walk above; their inner ifs are not tracked separately,
        # which only makes the check stricter for calls found there

    for stmt in fn.body:
        walk_stmt(stmt, [])
    return out


def test_step_waits_for_the_frame_path_and_fetches_the_poses_unconditionally():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    steps = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "step"]
    assert len(steps) == 1
    calls = _calls_with_conditions(steps[0])
    for name in ("ctx.sync", "po.fetch"):
        hits = [c for c in calls if c[0] == name]
        assert hits, name + " is not called in step()"
        free = [c for c in hits if not any("step_trace" in t or "environ" in t for t in c[1])]
        assert free, "%s only runs under a developer switch: %s" % (name, hits)
        # the only condition allowed around them is the developer split of the step into its halves
        for c in free:
            assert all(t in ('part != "ba"', "part != 'ba'") for t in c[1]), (name, c[1])
    # the streaming pass's upload of the NEXT step's frames: conditioned on `streaming` alone (it sat under the step-trace switch until round 6 -- the
    # default run's with_input_streaming figure then contained one upload, not one per step)
    ups = [c for c in calls if c[0] == "ctx.upload_async"]
    assert ups, "ctx.upload_async is not called in step()"
    assert any(c[1] == ["streaming"] for c in ups), ups
    # the gather of the trajectory uses what the fetch returned
    src = ast.unparse(steps[0])
    assert "frame_poses" in src and src.index("po.fetch") < src.index("cdist.gather_trajectory")
