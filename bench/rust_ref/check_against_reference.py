#!/usr/bin/env python
"""bench/rust_ref/check_against_reference.py [/root/reference] — does src/main.rs still name things salva3d has?

The pinning kit has never met cargo (no Rust toolchain where it was written).  This script is the part of `cargo check` that can be
done with a text search: every `use salva3d::path::{Items}` of main.rs must resolve to a `pub mod` chain under the reference's
src/ (lib.rs:86-118) ending in a `pub struct / enum / trait / type / fn / use ... Item`, and every `LiquidWorld` / `Fluid` /
`Boundary` method the runner calls must exist as `pub fn` in the file that defines the type.  Run by tests/test_host_logic.py when
the reference tree is present (it is not on the GPU boxes)."""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def uses(text):
    """[(module path list, item)] of every `use salva3d::...;` statement."""
    out = []
    for m in re.finditer(r"use\s+salva3d::([^;]+);", text, re.S):
        body = re.sub(r"\s+", "", m.group(1))
        if "{" in body:
            head, items = body.split("{", 1)
            items = items.rstrip("}").rstrip(",").split(",")
            path = [p for p in head.rstrip(":").split("::") if p]
        else:
            *path, item = body.split("::")
            items = [item]
        out.extend((path, it) for it in items if it)
    return out


def module_file(src, path):
    """The file that holds module `path` (a/b.rs or a/b/mod.rs); None if the chain is not all `pub mod`."""
    cur_dir, cur_file = src, os.path.join(src, "lib.rs")
    for name in path:
        text = open(cur_file).read()
        if not re.search(r"^\s*pub\s+mod\s+%s\s*;" % re.escape(name), text, re.M):
            # (re-exported into the parent? `pub use self::name::*` counts as reachable through the parent, not as a module)
            return None
        base = cur_dir if os.path.basename(cur_file) in ("lib.rs", "mod.rs") else os.path.splitext(cur_file)[0]
        for cand in (os.path.join(base, name + ".rs"), os.path.join(base, name, "mod.rs")):
            if os.path.exists(cand):
                cur_file, cur_dir = cand, os.path.dirname(cand)
                break
        else:
            return None
    return cur_file


def defines(file, item, src):
    text = open(file).read()
    if re.search(r"pub\s+(struct|enum|trait|type|fn|const)\s+%s\b" % re.escape(item), text):
        return True
    # pub use self::sub::{.. Item ..} / pub use self::sub::*  -> look into the sub-modules next to the file
    base = os.path.dirname(file) if os.path.basename(file) in ("lib.rs", "mod.rs") else os.path.splitext(file)[0]
    for m in re.finditer(r"pub\s+use\s+(?:self::|crate::)?([\w:]+)::(\{[^}]*\}|\*|\w+)\s*;", text, re.S):
        names = m.group(2)
        if names != "*" and not re.search(r"\b%s\b" % re.escape(item), names):
            continue
        parts = m.group(1).split("::")
        for root in (base, os.path.join(src)):
            for cand in (os.path.join(root, *parts) + ".rs", os.path.join(root, *parts, "mod.rs")):
                if os.path.exists(cand) and defines(cand, item, src):
                    return True
    return False


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = os.path.join(ref, "src")
    text = open(os.path.join(HERE, "src", "main.rs")).read()
    bad = []
    for path, item in uses(text):
        f = module_file(src, path) if path else os.path.join(src, "lib.rs")
        if f is None or not defines(f, item, src):
            bad.append("salva3d::" + "::".join(path + [item]))
    methods = {"liquid_world.rs": ["new", "step", "add_fluid", "add_boundary", "fluids", "boundaries"],
               "object/fluid.rs": ["new", "num_particles"], "object/boundary.rs": ["new"]}
    for rel, names in methods.items():
        t = open(os.path.join(src, rel)).read()
        for n in names:
            if re.search(r"\.%s\(|::%s\(" % (n, n), text) and not re.search(r"pub\s+fn\s+%s\b" % n, t):
                bad.append(f"{rel}: pub fn {n}")
    if bad:
        print("bench/rust_ref/src/main.rs names things the reference does not have:\n  " + "\n  ".join(bad))
        return 1
    print(f"bench/rust_ref/src/main.rs: {len(uses(text))} imported items and the methods it calls exist in {src}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
