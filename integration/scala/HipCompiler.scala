// integration/scala/HipCompiler.scala -- NOT BUILT IN THIS REPOSITORY (no JDK / sbt in the build image).
// A Rainier maintainer drops this file into rainier-compute/src/main/scala/com/stripe/rainier/compute/ (it needs the
// package-private Translator).  It replaces Compiler.compileTargets (compute/Compiler.scala:14-20) by a serialiser:
// the very inputs the ASM back end receives -- inputs: Seq[ir.Param], exprs: Seq[(String, Expr)]
// (compute/Compiler.scala:22-30) -- are written as RIR (include/rainier_hip_rir.h) for rh_model_create.
package com.stripe.rainier.compute

import com.stripe.rainier.ir._
import java.io.ByteArrayOutputStream
import java.nio.{ByteBuffer, ByteOrder}
import scala.collection.mutable

object HipCompiler {
  final case class Rir(bytes: Array[Byte], columns: Array[Array[Double]], rows: Array[Long], nParams: Int)

  private val binaryOp: Map[BinaryOp, Int] =
    Map(AddOp -> 2, SubtractOp -> 3, MultiplyOp -> 4, DivideOp -> 5, PowOp -> 6, CompareOp -> 7)
  private val unaryOp: Map[UnaryOp, Int] =
    Map(ExpOp -> 8, LogOp -> 9, AbsOp -> 10, NoOp -> 11, SinOp -> 12, CosOp -> 13, TanOp -> 14,
        AsinOp -> 15, AcosOp -> 16, AtanOp -> 17)

  /** kind 0: a density program (prior + one target per likelihood, outputs = value :: gradient). */
  def compileTargets(group: TargetGroup): Rir = {
    val nParams = group.parameters.size
    val perTarget = 1 + nParams                                        // compute/Target.scala:50-56
    val nTargets = group.data.length
    require(group.outputs.size == nTargets * perTarget)
    val nCols = group.data.map(_.length)
    val bytes = write(group.inputs, group.outputs.map(_._2), nParams, nCols, perTarget, kind = 0)
    Rir(bytes, group.data.flatten, group.data.map(cs => if (cs.isEmpty) 0L else cs.head.length.toLong), nParams)
  }

  /** kind 1: a requirements program for rh_requirements_eval (Generator.prepare, core/Generator.scala:76-84):
    * one data-free target per requirement, outputs(0) = the requirement, gradient slots = constant 0. */
  def compileRequirements(parameters: Seq[Parameter], reqs: Seq[Real]): Array[Byte] = {
    val zero: Real = Real.zero
    val outs = reqs.flatMap(r => r +: Seq.fill(parameters.size)(zero))
    write(parameters.map(_.param), outs, parameters.size, Array.fill(reqs.size)(0), 1 + parameters.size, kind = 1)
  }

  private def write(inputs: Seq[Param], outputs: Seq[Real], nParams: Int, nCols: Array[Int], perTarget: Int, kind: Int): Array[Byte] = {
    val translator = new Translator
    val exprs = outputs.map(translator.toExpr)                         // VarDef at first use, VarRef afterwards
    val inputIndex = new java.util.IdentityHashMap[Param, Integer]
    inputs.zipWithIndex.foreach { case (p, i) => inputIndex.put(p, i) }
    val ids = mutable.Map.empty[Sym, Int]
    val nodes = new ByteArrayOutputStream
    var n = 0
    def u32(xs: Int*): Unit = xs.foreach { x => nodes.write(ByteBuffer.allocate(4).order(ByteOrder.LITTLE_ENDIAN).putInt(x).array) }
    def f64(v: Double): Unit = nodes.write(ByteBuffer.allocate(8).order(ByteOrder.LITTLE_ENDIAN).putDouble(v).array)
    def fresh(): Int = { n += 1; n - 1 }

    def ref(e: Expr): Int = e match {
      case Const(v) =>
        require(!v.isNaN, "NaN constant")                              // compute/ToReal.scala:16-17
        u32(0); f64(v); fresh()
      case p: Param    => u32(1, inputIndex.get(p)); fresh()
      case VarRef(sym) => ids(sym)
      case VarDef(sym, rhs) =>
        val id = rhs match {
          case BinaryIR(l, r, op) => val a = ref(l); val b = ref(r); u32(binaryOp(op), a, b); fresh()
          case UnaryIR(x, op)     => val a = ref(x); u32(unaryOp(op), a); fresh()
          case LookupIR(i, table, low) =>
            val a = ref(i); val ts = table.map(ref)
            u32(18, a, low, ts.size); u32(ts: _*); fresh()
          case SeqIR(first, second) => val a = ref(first); val b = ref(second); u32(19, a, b); fresh()
          case MethodRef(_)         => sys.error("MethodRef is packer-internal (ir/Packer.scala:62-63)")
        }
        ids(sym) = id
        id
    }
    val outIds = exprs.map(ref)

    val out = new ByteArrayOutputStream
    def h32(xs: Int*): Unit = xs.foreach { x => out.write(ByteBuffer.allocate(4).order(ByteOrder.LITTLE_ENDIAN).putInt(x).array) }
    h32(0x31524952, 1, nParams, nCols.length, n, kind)                 // "RIR1", version, n_params, n_targets, n_nodes, kind
    nCols.indices.foreach { t =>
      h32(nCols(t), 0)
      h32(outIds.slice(t * perTarget, (t + 1) * perTarget): _*)
    }
    out.write(nodes.toByteArray)
    out.toByteArray
  }
}
