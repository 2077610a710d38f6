/* TEST INFRASTRUCTURE ONLY — a minimal stand-in for the JDK's <jni.h>, just enough for `cc -fsyntax-only` to type-check
 * jni/dismember_jni.c in an image without a JDK (tests/test_jni_shim.py).  It declares the JNI types and the handful of
 * JNIEnv functions the shim calls, with the JDK's signatures.  It is never linked or shipped; a real build uses
 * $JAVA_HOME/include/jni.h (jni/Makefile). */
#ifndef DM_TEST_JNI_STUB_H
#define DM_TEST_JNI_STUB_H
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef uint8_t jboolean; typedef float jfloat; typedef double jdouble;
typedef jint jsize;
struct _jobject; typedef struct _jobject *jobject;
typedef jobject jclass, jstring, jarray, jthrowable, jobjectArray, jintArray, jlongArray, jbyteArray, jfloatArray, jdoubleArray;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv *, const char *);
  jint (*ThrowNew)(JNIEnv *, jclass, const char *);
  jstring (*NewStringUTF)(JNIEnv *, const char *);
  const char *(*GetStringUTFChars)(JNIEnv *, jstring, jboolean *);
  void (*ReleaseStringUTFChars)(JNIEnv *, jstring, const char *);
  jobject (*GetObjectArrayElement)(JNIEnv *, jobjectArray, jsize);
  jsize (*GetArrayLength)(JNIEnv *, jarray);
  jdouble *(*GetDoubleArrayElements)(JNIEnv *, jdoubleArray, jboolean *);
  void (*ReleaseDoubleArrayElements)(JNIEnv *, jdoubleArray, jdouble *, jint);
  jfloat *(*GetFloatArrayElements)(JNIEnv *, jfloatArray, jboolean *);
  void (*ReleaseFloatArrayElements)(JNIEnv *, jfloatArray, jfloat *, jint);
  jint *(*GetIntArrayElements)(JNIEnv *, jintArray, jboolean *);
  void (*ReleaseIntArrayElements)(JNIEnv *, jintArray, jint *, jint);
  jlong *(*GetLongArrayElements)(JNIEnv *, jlongArray, jboolean *);
  void (*ReleaseLongArrayElements)(JNIEnv *, jlongArray, jlong *, jint);
  jbyte *(*GetByteArrayElements)(JNIEnv *, jbyteArray, jboolean *);
  void (*ReleaseByteArrayElements)(JNIEnv *, jbyteArray, jbyte *, jint);
  void *(*GetPrimitiveArrayCritical)(JNIEnv *, jarray, jboolean *);
  void (*ReleasePrimitiveArrayCritical)(JNIEnv *, jarray, void *, jint);
};
#endif
