/* vmnjni_rs.c -- see vmnjni_rs.h.  Plain C; needs a JDK's jni.h (not available where this repository is developed: there it is
 * type-checked against tests/jni_stub/jni.h, a syntax stand-in, by tests/test_jni_binding.py). */
#include "vmnjni_rs.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

struct vmn_jrs {
    JavaVM* vm;
    jobject bridge;                 /* global reference */
    jmethodID ring, ints, seed;
    jbyteArray rows;                /* global reference to the rows handed out last, pinned through rows_ptr */
    jbyte* rows_ptr;
    size_t row_bytes;               /* width of one row the library reads (vmn_group_exp_bytes of the group the source serves) */
    void* owner;
    struct vmn_jrs* next;           /* owner table */
};

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static struct vmn_jrs* g_owned = NULL;

static JNIEnv* env_of(struct vmn_jrs* h) {
    JNIEnv* env = NULL;
    if ((*h->vm)->GetEnv(h->vm, (void**)&env, JNI_VERSION_1_6) != JNI_OK) return NULL;   /* callbacks come on a Java thread */
    return env;
}
static void drop_rows(JNIEnv* env, struct vmn_jrs* h) {
    if (h->rows) {
        (*env)->ReleaseByteArrayElements(env, h->rows, h->rows_ptr, JNI_ABORT);
        (*env)->DeleteGlobalRef(env, h->rows);
        h->rows = NULL;
        h->rows_ptr = NULL;
    }
}
/* keep `arr` (n rows) alive and pinned; returns 0 on success.  A block shorter than n rows is refused: the library would
 * read n * row_bytes bytes from it. */
static int hand_over(JNIEnv* env, struct vmn_jrs* h, jbyteArray arr, size_t n, const uint8_t** rows) {
    if ((*env)->ExceptionCheck(env) || !arr) return 1;          /* the Java exception stays pending and surfaces in the caller */
    if ((size_t)(*env)->GetArrayLength(env, arr) < n * h->row_bytes) {
        (*env)->DeleteLocalRef(env, arr);
        return 1;
    }
    drop_rows(env, h);
    h->rows = (jbyteArray)(*env)->NewGlobalRef(env, arr);
    (*env)->DeleteLocalRef(env, arr);
    if (!h->rows) return 1;
    h->rows_ptr = (*env)->GetByteArrayElements(env, h->rows, NULL);
    if (!h->rows_ptr) return 1;
    *rows = (const uint8_t*)h->rows_ptr;
    return 0;
}
static int cb_ring(void* user, size_t n, const uint8_t** rows) {
    struct vmn_jrs* h = (struct vmn_jrs*)user;
    JNIEnv* env = env_of(h);
    if (!env) return 1;
    return hand_over(env, h, (jbyteArray)(*env)->CallObjectMethod(env, h->bridge, h->ring, (jlong)n), n, rows);
}
static int cb_ints(void* user, size_t n, int bits, const uint8_t** rows) {
    struct vmn_jrs* h = (struct vmn_jrs*)user;
    JNIEnv* env = env_of(h);
    if (!env) return 1;
    return hand_over(env, h, (jbyteArray)(*env)->CallObjectMethod(env, h->bridge, h->ints, (jlong)n, (jint)bits), n, rows);
}
static int cb_seed(void* user, uint8_t seed_out[32]) {
    struct vmn_jrs* h = (struct vmn_jrs*)user;
    JNIEnv* env = env_of(h);
    if (!env) return 1;
    jbyteArray arr = (jbyteArray)(*env)->CallObjectMethod(env, h->bridge, h->seed);
    if ((*env)->ExceptionCheck(env) || !arr || (*env)->GetArrayLength(env, arr) != 32) return 1;
    (*env)->GetByteArrayRegion(env, arr, 0, 32, (jbyte*)seed_out);
    (*env)->DeleteLocalRef(env, arr);
    return 0;
}

vmn_jrs* vmn_jrs_new(JNIEnv* env, jobject bridge, size_t row_bytes) {
    struct vmn_jrs* h = (struct vmn_jrs*)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->row_bytes = row_bytes;
    jclass c = (*env)->GetObjectClass(env, bridge);
    h->ring = (*env)->GetMethodID(env, c, "ringElements", "(J)[B");
    h->ints = (*env)->GetMethodID(env, c, "integers", "(JI)[B");
    h->seed = (*env)->GetMethodID(env, c, "arraySeed", "()[B");
    if (!h->ring || !h->ints || !h->seed || (*env)->GetJavaVM(env, &h->vm) != JNI_OK) {
        free(h);
        return NULL;
    }
    h->bridge = (*env)->NewGlobalRef(env, bridge);
    /* does this source offer seeds?  RandomSourceBridge.deviceArrays() says so without drawing randomness */
    jmethodID dev = (*env)->GetMethodID(env, c, "deviceArrays", "()Z");
    if (dev && !(*env)->CallBooleanMethod(env, bridge, dev)) h->seed = NULL;
    return h;
}
void vmn_jrs_fill(vmn_jrs* h, vmn_random_source* out) {
    out->user = h;
    out->ring_elements = cb_ring;
    out->integers = cb_ints;
    out->array_seed = h->seed ? cb_seed : NULL;
}
void vmn_jrs_free(JNIEnv* env, vmn_jrs* h) {
    if (!h) return;
    drop_rows(env, h);
    if (h->bridge) (*env)->DeleteGlobalRef(env, h->bridge);
    free(h);
}
void vmn_jrs_set_owner(vmn_jrs* h, void* owner) {
    h->owner = owner;
    pthread_mutex_lock(&g_mu);
    h->next = g_owned;
    g_owned = h;
    pthread_mutex_unlock(&g_mu);
}
struct vmn_jcomm {
    JavaVM* vm;
    jobject bridge;
    jmethodID gather;
    int rank, world;
    void* owner;
    struct vmn_jcomm* next;
};
static struct vmn_jcomm* g_comms = NULL;

static int cb_gather(void* user, const uint8_t* send, size_t bytes, uint8_t* recv) {
    struct vmn_jcomm* h = (struct vmn_jcomm*)user;
    JNIEnv* env = NULL;
    if ((*h->vm)->GetEnv(h->vm, (void**)&env, JNI_VERSION_1_6) != JNI_OK) return 1;
    jbyteArray mine = (*env)->NewByteArray(env, (jsize)bytes);
    if (!mine) return 1;
    (*env)->SetByteArrayRegion(env, mine, 0, (jsize)bytes, (const jbyte*)send);
    jbyteArray all = (jbyteArray)(*env)->CallObjectMethod(env, h->bridge, h->gather, mine);
    (*env)->DeleteLocalRef(env, mine);
    if ((*env)->ExceptionCheck(env) || !all || (size_t)(*env)->GetArrayLength(env, all) != bytes * (size_t)h->world) return 1;
    (*env)->GetByteArrayRegion(env, all, 0, (jsize)(bytes * (size_t)h->world), (jbyte*)recv);
    (*env)->DeleteLocalRef(env, all);
    return 0;
}
vmn_jcomm* vmn_jcomm_new(JNIEnv* env, jobject bridge, void* owner) {
    struct vmn_jcomm* h = (struct vmn_jcomm*)calloc(1, sizeof(*h));
    if (!h) return NULL;
    jclass c = (*env)->GetObjectClass(env, bridge);
    jmethodID rank = (*env)->GetMethodID(env, c, "rank", "()I"), world = (*env)->GetMethodID(env, c, "world", "()I");
    h->gather = (*env)->GetMethodID(env, c, "allGather", "([B)[B");
    if (!rank || !world || !h->gather || (*env)->GetJavaVM(env, &h->vm) != JNI_OK) {
        free(h);
        return NULL;
    }
    h->rank = (*env)->CallIntMethod(env, bridge, rank);
    h->world = (*env)->CallIntMethod(env, bridge, world);
    h->bridge = (*env)->NewGlobalRef(env, bridge);
    h->owner = owner;
    pthread_mutex_lock(&g_mu);
    h->next = g_comms;
    g_comms = h;
    pthread_mutex_unlock(&g_mu);
    return h;
}
void vmn_jcomm_fill(vmn_jcomm* h, vmn_comm* out) {
    out->user = h;
    out->rank = h->rank;
    out->world = h->world;
    out->all_gather = cb_gather;
}
static void release_comms(void* owner) {
    struct vmn_jcomm* found = NULL;
    pthread_mutex_lock(&g_mu);
    for (struct vmn_jcomm** pp = &g_comms; *pp; pp = &(*pp)->next) {
        if ((*pp)->owner == owner) {
            found = *pp;
            *pp = found->next;
            break;
        }
    }
    pthread_mutex_unlock(&g_mu);
    if (found) {
        JNIEnv* env = NULL;
        if ((*found->vm)->GetEnv(found->vm, (void**)&env, JNI_VERSION_1_6) == JNI_OK) (*env)->DeleteGlobalRef(env, found->bridge);
        free(found);
    }
}

void vmn_jrs_release_owner(void* owner) {
    release_comms(owner);
    struct vmn_jrs* found = NULL;
    pthread_mutex_lock(&g_mu);
    for (struct vmn_jrs** pp = &g_owned; *pp; pp = &(*pp)->next) {
        if ((*pp)->owner == owner) {
            found = *pp;
            *pp = found->next;
            break;
        }
    }
    pthread_mutex_unlock(&g_mu);
    if (found) {
        JNIEnv* env = env_of(found);
        if (env) vmn_jrs_free(env, found);
    }
}
