/* tests/stubs/jni.h -- NOT the JDK's jni.h: a stand-in with exactly the JNI entry points that
 * rainier_amd/jni/rainier_hip_jni.c uses, so that the shim can be compiled in an image without a JDK and executed against
 * tests/stubs/fake_jni.c (tests/test_jni_shim.py).  Signatures follow the JNI specification (jni.h, JNINativeInterface_);
 * the real table has ~230 slots, the shim only ever reaches these through the JNIEnv, so the layout difference is moot here. */
#ifndef RH_STUB_JNI_H
#define RH_STUB_JNI_H
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef double jdouble; typedef jint jsize;
typedef unsigned char jboolean;
typedef struct _jobject *jobject;
typedef jobject jclass, jarray, jobjectArray, jbyteArray, jintArray, jlongArray, jdoubleArray, jthrowable;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_COMMIT 1
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv *, const char *);
  jint (*ThrowNew)(JNIEnv *, jclass, const char *);
  jsize (*GetArrayLength)(JNIEnv *, jarray);
  jobject (*GetObjectArrayElement)(JNIEnv *, jobjectArray, jsize);
  jbyte *(*GetByteArrayElements)(JNIEnv *, jbyteArray, jboolean *);
  jint *(*GetIntArrayElements)(JNIEnv *, jintArray, jboolean *);
  jlong *(*GetLongArrayElements)(JNIEnv *, jlongArray, jboolean *);
  jdouble *(*GetDoubleArrayElements)(JNIEnv *, jdoubleArray, jboolean *);
  void (*ReleaseByteArrayElements)(JNIEnv *, jbyteArray, jbyte *, jint);
  void (*ReleaseIntArrayElements)(JNIEnv *, jintArray, jint *, jint);
  void (*ReleaseLongArrayElements)(JNIEnv *, jlongArray, jlong *, jint);
  void (*ReleaseDoubleArrayElements)(JNIEnv *, jdoubleArray, jdouble *, jint);
};
#endif
