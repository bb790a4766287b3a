"""Minimal model front-end that produces RIR (the engine's input language).

In a Rainier deployment the JVM side serialises what `Compiler.compileTargets` hands to
`ir.CompiledFunction` (rainier-compute/.../compute/Compiler.scala:14-30) -- see INTEGRATION.md.
There is no JVM in this environment, so tests and bench.py author models with this small
expression DSL instead.  It mirrors, in miniature, the three reference pieces that sit *before*
the hot path:

* hash-consed expression building        (compute/Translator.scala:154-187 `SymCache`)
* symbolic reverse-mode differentiation  (compute/Gradient.scala:8-153; same local rules,
  incl. `abs` -> eq(x,0,0,g*x/|x|) and Lookup -> eq(index,k,g,0))
* one target per likelihood, outputs `[value, grad_0 .. grad_{n-1}]`
  (compute/Target.scala:41-56)

It does NOT reproduce the reference's algebraic normal forms (Line/LogLine) or its partial
evaluation of data (`PartialEvaluator.inline`, compute/Target.scala:20-24): row expressions are
deliberately left un-inlined so that the engine streams the observation columns.

The byte layout written by `Graph.compile` is documented in include/rainier_hip_rir.h.
"""
from __future__ import annotations

import math
import struct
from typing import Dict, List, Sequence, Tuple

# opcodes (ir/IR.scala:3-41, ir/Ops.scala:3-37)
CONST, INPUT = 0, 1
ADD, SUB, MUL, DIV, POW, COMPARE = 2, 3, 4, 5, 6, 7
EXP, LOG, ABS, NOOP, SIN, COS, TAN, ASIN, ACOS, ATAN = 8, 9, 10, 11, 12, 13, 14, 15, 16, 17
LOOKUP, SEQ = 18, 19
_BINARY = (ADD, SUB, MUL, DIV, POW, COMPARE)
_UNARY = (EXP, LOG, ABS, NOOP, SIN, COS, TAN, ASIN, ACOS, ATAN)
_COMMUTATIVE = (ADD, MUL)
RIR_MAGIC = 0x31524952  # "RIR1"


class Expr:
    """A handle on one node of a Graph (operator overloading sugar)."""

    __slots__ = ("g", "id")

    def __init__(self, g: "Graph", id: int):
        self.g = g
        self.id = id

    def _w(self, o) -> "Expr":
        return o if isinstance(o, Expr) else self.g.const(float(o))

    def __add__(self, o): return self.g.binary(ADD, self, self._w(o))
    def __radd__(self, o): return self.g.binary(ADD, self._w(o), self)
    def __sub__(self, o): return self.g.binary(SUB, self, self._w(o))
    def __rsub__(self, o): return self.g.binary(SUB, self._w(o), self)
    def __mul__(self, o): return self.g.binary(MUL, self, self._w(o))
    def __rmul__(self, o): return self.g.binary(MUL, self._w(o), self)
    def __truediv__(self, o): return self.g.binary(DIV, self, self._w(o))
    def __rtruediv__(self, o): return self.g.binary(DIV, self._w(o), self)
    def __neg__(self): return self.g.binary(SUB, self.g.const(0.0), self)
    def __pow__(self, o): return self.g.binary(POW, self, self._w(o))
    def exp(self): return self.g.unary(EXP, self)
    def log(self): return self.g.unary(LOG, self)
    def abs(self): return self.g.unary(ABS, self)
    def sin(self): return self.g.unary(SIN, self)
    def cos(self): return self.g.unary(COS, self)
    def tan(self): return self.g.unary(TAN, self)
    def asin(self): return self.g.unary(ASIN, self)
    def acos(self): return self.g.unary(ACOS, self)
    def atan(self): return self.g.unary(ATAN, self)
    def compare(self, o): return self.g.binary(COMPARE, self, self._w(o))


class Graph:
    """Hash-consed expression DAG over `n_params` parameters and per-target data columns.

    `target_cols[t]` = number of data columns of target t (0 for the prior / data-free targets).
    Input numbering is the DataFunction layout (ir/DataFunction.scala:3-12): parameters first,
    then each target's columns.
    """

    def __init__(self, n_params: int, target_cols: Sequence[int]):
        self.n_params = int(n_params)
        self.target_cols = [int(c) for c in target_cols]
        self.nodes: List[tuple] = []
        self._cache: Dict[tuple, int] = {}
        self._col_start = []
        s = self.n_params
        for c in self.target_cols:
            self._col_start.append(s)
            s += c
        self.n_inputs = s

    # -- node construction ---------------------------------------------------------------
    def _mk(self, key: tuple) -> Expr:
        i = self._cache.get(key)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(key)
            self._cache[key] = i
        return Expr(self, i)

    def const(self, v: float) -> Expr:
        v = float(v)
        if math.isnan(v):
            raise ArithmeticError("NaN constant")  # compute/ToReal.scala:16-17
        return self._mk((CONST, struct.pack("<d", v)))

    def param(self, i: int) -> Expr:
        assert 0 <= i < self.n_params
        return self._mk((INPUT, i))

    def col(self, target: int, j: int) -> Expr:
        assert 0 <= j < self.target_cols[target]
        return self._mk((INPUT, self._col_start[target] + j))

    def _cval(self, e: Expr):
        k = self.nodes[e.id]
        return struct.unpack("<d", k[1])[0] if k[0] == CONST else None

    def binary(self, op: int, a: Expr, b: Expr) -> Expr:
        ca, cb = self._cval(a), self._cval(b)
        # light constant folding, in the spirit of compute/RealOps.scala (identities only)
        if ca is not None and cb is not None and op in (ADD, SUB, MUL, DIV):
            try:
                v = {ADD: ca + cb, SUB: ca - cb, MUL: ca * cb}[op] if op != DIV else ca / cb
                if not math.isnan(v):
                    return self.const(v)
            except ZeroDivisionError:
                pass
        if op == ADD:
            if ca == 0.0: return b
            if cb == 0.0: return a
        if op == SUB and cb == 0.0: return a
        if op == MUL:
            if ca == 1.0: return b
            if cb == 1.0: return a
            if ca == 0.0 or cb == 0.0: return self.const(0.0)
        if op == DIV and cb == 1.0: return a
        if op == POW:
            if cb == 1.0: return a
            if cb == 2.0: return self.binary(MUL, a, a)  # compute/Translator.scala:101-114 x^2 -> x*x
            if cb == 0.0: return self.const(1.0)
        key = (op, a.id, b.id)
        if op in _COMMUTATIVE and key not in self._cache and (op, b.id, a.id) in self._cache:
            key = (op, b.id, a.id)
        return self._mk(key)

    def unary(self, op: int, a: Expr) -> Expr:
        return self._mk((op, a.id))

    def lookup(self, index: Expr, table: Sequence[Expr], low: int = 0) -> Expr:
        return self._mk((LOOKUP, index.id, int(low), tuple(t.id for t in table)))

    def eq(self, a, b, if_true, if_false) -> Expr:
        """Real.eq (compute/Real.scala:83-99): Lookup(Compare(a,b), [F, T, F], low=-1)."""
        a, b = self._w(a), self._w(b)
        t, f = self._w(if_true), self._w(if_false)
        return self.lookup(a.compare(b), [f, t, f], -1)

    def _w(self, o) -> Expr:
        return o if isinstance(o, Expr) else self.const(float(o))

    def sum(self, xs: Sequence[Expr]) -> Expr:
        xs = list(xs)
        if not xs:
            return self.const(0.0)
        acc = xs[0]
        for x in xs[1:]:
            acc = acc + x  # left fold, as compute/Translator.scala:116-125 does for Line
        return acc

    # -- symbolic reverse-mode differentiation (compute/Gradient.scala) -------------------
    def gradient(self, root: Expr) -> List[Expr]:
        zero, one = self.const(0.0), self.const(1.0)
        adj: Dict[int, Expr] = {root.id: one}

        def acc(i: int, e: Expr):
            adj[i] = e if i not in adj else adj[i] + e

        # nodes are created operands-first, so descending id is a valid reverse topological order
        reach = self._reachable([root.id])
        for i in sorted(reach, reverse=True):
            g = adj.get(i)
            if g is None:
                continue
            k = self.nodes[i]
            op = k[0]
            if op in (CONST, INPUT):
                continue
            if op in _BINARY:
                a, b = Expr(self, k[1]), Expr(self, k[2])
                if op == ADD: acc(a.id, g); acc(b.id, g)
                elif op == SUB: acc(a.id, g); acc(b.id, zero - g)
                elif op == MUL: acc(a.id, g * b); acc(b.id, g * a)
                elif op == DIV:
                    acc(a.id, g / b)
                    acc(b.id, zero - g * a / (b * b))
                elif op == POW:
                    me = Expr(self, i)
                    acc(a.id, g * b * (a ** (b - 1.0)))
                    if self._depends_on_input(b.id):
                        acc(b.id, g * me * self.eq(a, zero, one, a).log())
                # COMPARE: piecewise constant, no gradient
            elif op in _UNARY:
                a = Expr(self, k[1]); me = Expr(self, i)
                if op == LOG: acc(a.id, g * (one / a))
                elif op == EXP: acc(a.id, g * me)
                elif op == ABS: acc(a.id, self.eq(a, zero, zero, g * a / me))
                elif op == NOOP: acc(a.id, g)
                elif op == SIN: acc(a.id, g * a.cos())
                elif op == COS: acc(a.id, g * (zero - a.sin()))
                elif op == TAN: acc(a.id, g / (a.cos() ** 2.0))
                elif op == ASIN: acc(a.id, g / ((one - a ** 2.0) ** 0.5))
                elif op == ACOS: acc(a.id, (zero - g) / ((one - a ** 2.0) ** 0.5))
                elif op == ATAN: acc(a.id, g / (one + a ** 2.0))
            elif op == LOOKUP:
                index = Expr(self, k[1]); low = k[2]
                for j, t in enumerate(k[3]):
                    acc(t, self.eq(index, float(low + j), g, zero))
            elif op == SEQ:
                acc(k[2], g)
        return [adj.get(self.param(p).id, zero) for p in range(self.n_params)]

    def _depends_on_input(self, i: int) -> bool:
        return any(self.nodes[j][0] == INPUT for j in self._reachable([i]))

    def _operands(self, i: int) -> Tuple[int, ...]:
        k = self.nodes[i]
        op = k[0]
        if op in (CONST, INPUT): return ()
        if op in _BINARY or op == SEQ: return (k[1], k[2])
        if op in _UNARY: return (k[1],)
        if op == LOOKUP: return (k[1],) + tuple(k[3])
        raise ValueError(op)

    def _reachable(self, roots: Sequence[int]) -> set:
        seen = set()
        stack = list(roots)
        while stack:
            i = stack.pop()
            if i in seen:
                continue
            seen.add(i)
            stack.extend(self._operands(i))
        return seen

    # -- serialisation -------------------------------------------------------------------
    def compile_requirements(self, exprs: Sequence[Expr]) -> bytes:
        """Requirements program (header kind 1) for rh_requirements_eval: what Generator.prepare compiles with
        Compiler.default.compile(parameters, namedReqs) (core/Generator.scala:76-78).  Parameter-only expressions."""
        assert all(c == 0 for c in self.target_cols), "requirements cannot read data columns"
        zero = self.const(0.0)
        saved = self.target_cols
        self.target_cols = [0] * len(exprs)
        try:
            blob = bytearray(self.compile(list(exprs), gradients=[[zero] * self.n_params for _ in exprs]))
        finally:
            self.target_cols = saved
        blob[20:24] = struct.pack("<I", 1)
        return bytes(blob)

    def compile(self, targets: Sequence[Expr], gradients: Sequence[Sequence[Expr]] = None) -> bytes:
        """targets[t] = log-density contribution of target t (per row if it has columns).
        Returns the RIR blob with outputs [value, d/dθ_0 .. d/dθ_{n-1}] per target."""
        assert len(targets) == len(self.target_cols)
        outs: List[List[int]] = []
        for t, e in enumerate(targets):
            grads = self.gradient(e) if gradients is None else list(gradients[t])
            assert len(grads) == self.n_params
            outs.append([e.id] + [x.id for x in grads])
        keep = sorted(self._reachable([i for o in outs for i in o]))
        remap = {old: new for new, old in enumerate(keep)}
        w: List[bytes] = []
        u32 = lambda *xs: w.append(struct.pack("<%dI" % len(xs), *xs))
        u32(RIR_MAGIC, 1, self.n_params, len(targets), len(keep), 0)
        for t, o in enumerate(outs):
            u32(self.target_cols[t], 0)
            u32(*[remap[i] for i in o])
        for old in keep:
            k = self.nodes[old]
            op = k[0]
            if op == CONST:
                u32(op); w.append(k[1])
            elif op == INPUT:
                u32(op, k[1])
            elif op in _BINARY or op == SEQ:
                u32(op, remap[k[1]], remap[k[2]])
            elif op in _UNARY:
                u32(op, remap[k[1]])
            elif op == LOOKUP:
                u32(op, remap[k[1]])
                w.append(struct.pack("<i", k[2]))
                u32(len(k[3]), *[remap[i] for i in k[3]])
        return b"".join(w)
