/* tests/stubs/fake_jni.c -- TEST INFRASTRUCTURE: a JNIEnv function table without a JVM.
 *
 * Implements exactly the JNI calls rainier_amd/jni/rainier_hip_jni.c uses (tests/stubs/jni.h) over malloc'ed arrays,
 * with the semantics a copying JVM has, so that the shim can be EXECUTED in an image without a JDK:
 *   Get<T>ArrayElements  returns a fresh COPY of the array (isCopy = JNI_TRUE; HotSpot does this for non-pinned arrays);
 *   Release... mode 0    copies the buffer back into the array and frees it;
 *   Release... JNI_ABORT frees it without copying back -- results released with the wrong mode are therefore LOST,
 *                        inputs modified by mistake never reach the "Java" array;
 *   ThrowNew             records the pending exception (class name + message);
 * plus book-keeping of outstanding Get/Release pairs (a leak or a double release fails the test).
 * Python (tests/test_jni_shim.py) creates the arrays through the fj_* helpers and calls the shim's
 * Java_com_stripe_rainier_hip_Native_00024_* entry points through ctypes with fj_env().
 */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { FJ_BYTE = 1, FJ_INT = 2, FJ_LONG = 3, FJ_DOUBLE = 4, FJ_OBJECT = 5, FJ_CLASS = 6 };
struct _jobject {
  int kind;
  jsize len;
  void *data;         /* element storage (FJ_OBJECT: jobject[]) */
  char name[128];     /* FJ_CLASS */
};
static const size_t fj_elem[] = {0, 1, 4, 8, 8, sizeof(jobject), 0};

static int g_outstanding = 0, g_bad_release = 0, g_gets = 0, g_copybacks = 0;
static int g_pinning = 0;   /* 0: a copying JVM (Get returns a copy, isCopy = JNI_TRUE); 1: a pinning JVM (Get returns the array's own
                               storage, isCopy = JNI_FALSE, Release modes do not move data) -- the JNI specification allows both */
void fj_set_pinning(int on) { g_pinning = on; }
static char g_exc_class[128], g_exc_msg[1024];
static struct { void *buf; jobject arr; } g_live[256];

static jobject fj_alloc(int kind, jsize len) {
  jobject o = (jobject)calloc(1, sizeof(struct _jobject));
  o->kind = kind; o->len = len;
  o->data = calloc(len ? (size_t)len : 1, fj_elem[kind] ? fj_elem[kind] : 1);
  return o;
}
jobject fj_new_array(int kind, jsize len, const void *init) {
  jobject o = fj_alloc(kind, len);
  if (init && len) memcpy(o->data, init, (size_t)len * fj_elem[kind]);
  return o;
}
void fj_set_object(jobject arr, jsize i, jobject v) { ((jobject *)arr->data)[i] = v; }
void *fj_array_data(jobject arr) { return arr->data; }
jsize fj_array_len(jobject arr) { return arr->len; }
void fj_free(jobject o) { if (o) { free(o->data); free(o); } }
int fj_outstanding(void) { return g_outstanding; }
int fj_bad_releases(void) { return g_bad_release; }
int fj_gets(void) { return g_gets; }
int fj_copybacks(void) { return g_copybacks; }
const char *fj_exception_class(void) { return g_exc_class; }
const char *fj_exception_message(void) { return g_exc_msg; }
void fj_clear(void) { g_exc_class[0] = g_exc_msg[0] = 0; g_bad_release = 0; g_gets = 0; g_copybacks = 0; }

static jclass f_FindClass(JNIEnv *env, const char *name) {
  (void)env;
  jobject c = fj_alloc(FJ_CLASS, 0);
  snprintf(c->name, sizeof c->name, "%s", name);
  return c;   /* leaked on purpose: local references die with the native frame */
}
static jint f_ThrowNew(JNIEnv *env, jclass cls, const char *msg) {
  (void)env;
  snprintf(g_exc_class, sizeof g_exc_class, "%s", cls ? cls->name : "?");
  snprintf(g_exc_msg, sizeof g_exc_msg, "%s", msg ? msg : "");
  return 0;
}
static jsize f_GetArrayLength(JNIEnv *env, jarray a) { (void)env; return a->len; }
static jobject f_GetObjectArrayElement(JNIEnv *env, jobjectArray a, jsize i) {
  (void)env;
  if (a->kind != FJ_OBJECT || i < 0 || i >= a->len) { g_bad_release++; return NULL; }
  return ((jobject *)a->data)[i];
}
static void *get_elems(jarray a, int kind, jboolean *is_copy) {
  if (!a || a->kind != kind) { g_bad_release++; return NULL; }
  const size_t bytes = (size_t)a->len * fj_elem[kind];
  void *buf;
  if (g_pinning) { buf = a->data; if (is_copy) *is_copy = 0; }
  else { buf = malloc(bytes ? bytes : 1); memcpy(buf, a->data, bytes); if (is_copy) *is_copy = 1; }
  for (int i = 0; i < 256; i++) if (!g_live[i].buf) { g_live[i].buf = buf; g_live[i].arr = a; break; }
  g_outstanding++; g_gets++;
  return buf;
}
static void release_elems(jarray a, int kind, void *buf, jint mode) {
  int found = 0;
  for (int i = 0; i < 256; i++) if (g_live[i].buf == buf && buf) { found = g_live[i].arr == a; g_live[i].buf = NULL; break; }
  if (!found || !a || a->kind != kind) { g_bad_release++; return; }
  if (mode != 0 && mode != JNI_ABORT) g_bad_release++;   /* JNI_COMMIT is never right for the shim */
  if (buf != a->data) {
    if (mode == 0) { memcpy(a->data, buf, (size_t)a->len * fj_elem[kind]); g_copybacks++; }
    free(buf);
  }
  g_outstanding--;
}
static jbyte *f_GetByte(JNIEnv *e, jbyteArray a, jboolean *c) { (void)e; return (jbyte *)get_elems(a, FJ_BYTE, c); }
static jint *f_GetInt(JNIEnv *e, jintArray a, jboolean *c) { (void)e; return (jint *)get_elems(a, FJ_INT, c); }
static jlong *f_GetLong(JNIEnv *e, jlongArray a, jboolean *c) { (void)e; return (jlong *)get_elems(a, FJ_LONG, c); }
static jdouble *f_GetDouble(JNIEnv *e, jdoubleArray a, jboolean *c) { (void)e; return (jdouble *)get_elems(a, FJ_DOUBLE, c); }
static void f_RelByte(JNIEnv *e, jbyteArray a, jbyte *b, jint m) { (void)e; release_elems(a, FJ_BYTE, b, m); }
static void f_RelInt(JNIEnv *e, jintArray a, jint *b, jint m) { (void)e; release_elems(a, FJ_INT, b, m); }
static void f_RelLong(JNIEnv *e, jlongArray a, jlong *b, jint m) { (void)e; release_elems(a, FJ_LONG, b, m); }
static void f_RelDouble(JNIEnv *e, jdoubleArray a, jdouble *b, jint m) { (void)e; release_elems(a, FJ_DOUBLE, b, m); }

static const struct JNINativeInterface_ g_table = {
  f_FindClass, f_ThrowNew, f_GetArrayLength, f_GetObjectArrayElement, f_GetByte, f_GetInt, f_GetLong, f_GetDouble,
  f_RelByte, f_RelInt, f_RelLong, f_RelDouble,
};
static JNIEnv g_env = &g_table;
JNIEnv *fj_env(void) { return &g_env; }
