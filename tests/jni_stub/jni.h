/* tests/jni_stub/jni.h -- SYNTAX STAND-IN, test infrastructure only, never shipped and never linked.
 *
 * The development image has no JDK, so the JNI layer under jni/ could not be put through a compiler at all.  This file
 * declares just the types and the JNIEnv / JavaVM members that jni/*.c use, with the signatures of the JNI
 * specification (Java SE "JNI Functions" chapter), so that `gcc -fsyntax-only -Wall -Wextra -Werror` type-checks every
 * generated wrapper and the hand-written bridges (tests/test_jni_binding.py).  It is NOT a substitute for a JDK's jni.h:
 * nothing is implemented, nothing built with it can run, and a real build must use $JAVA_HOME/include/jni.h. */
#ifndef VMN_TEST_JNI_STUB_H
#define VMN_TEST_JNI_STUB_H
#include <stdarg.h>
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_OK 0
#define JNI_ABORT 2
#define JNI_COMMIT 1
#define JNI_VERSION_1_6 0x00010006

typedef int32_t jint;
typedef int64_t jlong;
typedef signed char jbyte;
typedef unsigned char jboolean;
typedef double jdouble;
typedef jint jsize;

struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;
struct _jmethodID;
typedef struct _jmethodID* jmethodID;

struct JNINativeInterface_;
struct JNIInvokeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
typedef const struct JNIInvokeInterface_* JavaVM;

struct JNINativeInterface_ {
    jclass (*GetObjectClass)(JNIEnv* env, jobject obj);
    jmethodID (*GetMethodID)(JNIEnv* env, jclass clazz, const char* name, const char* sig);
    jobject (*CallObjectMethod)(JNIEnv* env, jobject obj, jmethodID methodID, ...);
    jint (*CallIntMethod)(JNIEnv* env, jobject obj, jmethodID methodID, ...);
    jboolean (*CallBooleanMethod)(JNIEnv* env, jobject obj, jmethodID methodID, ...);
    jboolean (*ExceptionCheck)(JNIEnv* env);
    jobject (*NewGlobalRef)(JNIEnv* env, jobject lobj);
    void (*DeleteGlobalRef)(JNIEnv* env, jobject gref);
    void (*DeleteLocalRef)(JNIEnv* env, jobject obj);
    jstring (*NewStringUTF)(JNIEnv* env, const char* utf);
    const char* (*GetStringUTFChars)(JNIEnv* env, jstring str, jboolean* isCopy);
    void (*ReleaseStringUTFChars)(JNIEnv* env, jstring str, const char* chars);
    jsize (*GetArrayLength)(JNIEnv* env, jarray array);
    jbyteArray (*NewByteArray)(JNIEnv* env, jsize len);
    jbyte* (*GetByteArrayElements)(JNIEnv* env, jbyteArray array, jboolean* isCopy);
    jint* (*GetIntArrayElements)(JNIEnv* env, jintArray array, jboolean* isCopy);
    jlong* (*GetLongArrayElements)(JNIEnv* env, jlongArray array, jboolean* isCopy);
    jdouble* (*GetDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jboolean* isCopy);
    void (*ReleaseByteArrayElements)(JNIEnv* env, jbyteArray array, jbyte* elems, jint mode);
    void (*ReleaseIntArrayElements)(JNIEnv* env, jintArray array, jint* elems, jint mode);
    void (*ReleaseLongArrayElements)(JNIEnv* env, jlongArray array, jlong* elems, jint mode);
    void (*ReleaseDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jdouble* elems, jint mode);
    void (*GetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, jbyte* buf);
    void (*SetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, const jbyte* buf);
    void (*SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
    void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
    jlong (*GetDirectBufferCapacity)(JNIEnv* env, jobject buf);
    jint (*GetJavaVM)(JNIEnv* env, JavaVM** vm);
};

struct JNIInvokeInterface_ {
    jint (*GetEnv)(JavaVM* vm, void** penv, jint version);
};
#endif
