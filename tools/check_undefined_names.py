import ast, builtins, sys, glob
def names_defined(node):
    out=set()
    for n in ast.walk(node):
        if isinstance(n,(ast.Import,ast.ImportFrom)):
            for a in n.names: out.add((a.asname or a.name).split(".")[0])
        elif isinstance(n,(ast.FunctionDef,ast.ClassDef,ast.AsyncFunctionDef)): out.add(n.name)
        elif isinstance(n,ast.Name) and isinstance(n.ctx,(ast.Store,ast.Del)): out.add(n.id)
        elif isinstance(n,ast.arg): out.add(n.arg)
        elif isinstance(n,ast.ExceptHandler) and n.name: out.add(n.name)
    return out
bad=0
for f in sorted(glob.glob("tests/*.py")+glob.glob("llm_awq_amd/*.py")+["bench.py","__graft_entry__.py"]+glob.glob("tools/*.py")+glob.glob("oracle/*.py")):
    t=ast.parse(open(f).read())
    # module-level names: only those defined at module top level (not nested in functions)
    mod=set(dir(builtins))|{"__file__","__name__","__doc__"}
    for n in t.body:
        if isinstance(n,(ast.FunctionDef,ast.ClassDef,ast.AsyncFunctionDef)): mod.add(n.name)
        else: mod|=names_defined(n)
    def check(fn, scope):
        local=scope|names_defined(fn)
        for n in ast.walk(fn):
            if isinstance(n,ast.Name) and isinstance(n.ctx,ast.Load) and n.id not in local:
                print(f"{f}:{n.lineno}: undefined name {n.id}"); global bad; bad+=1
    for n in t.body:
        if isinstance(n,(ast.FunctionDef,ast.AsyncFunctionDef)): check(n, mod)
        elif isinstance(n,ast.ClassDef):
            for m in n.body:
                if isinstance(m,(ast.FunctionDef,ast.AsyncFunctionDef)): check(m, mod|{n.name})
print("undefined:",bad)
