/* vmnjni_rs.h -- bridge from vmn_random_source (include/vmnproofs.h) to a Java object implementing
 * com.verificatum.vmnhip.RandomSourceBridge:
 *     byte[] ringElements(long n)          n rows of exp_bytes, big-endian, each < q
 *     byte[] integers(long n, int bits)    n rows of exp_bytes holding `bits`-bit integers (as field elements)
 *     byte[] arraySeed()                   32 bytes for one device-expanded array draw, or null: host rows for arrays too
 * The callbacks run on the thread that called into the library (the party's protocol thread), so the JNIEnv of that
 * call is used.  Rows handed to the library stay pinned until the next callback or until the bridge is freed, as the C
 * ABI requires.  A proof object created with a bridge owns it: the table below maps the object's handle to the bridge,
 * and the object's _free wrapper releases it. */
#ifndef VMNJNI_RS_H
#define VMNJNI_RS_H
#include <jni.h>
#include "../include/vmnproofs.h"

typedef struct vmn_jrs vmn_jrs;
vmn_jrs* vmn_jrs_new(JNIEnv* env, jobject bridge, size_t row_bytes);   /* row_bytes = vmn_group_exp_bytes: every block handed back must hold n rows of it */
void vmn_jrs_fill(vmn_jrs* h, vmn_random_source* out);
void vmn_jrs_free(JNIEnv* env, vmn_jrs* h);
void vmn_jrs_set_owner(vmn_jrs* h, void* owner);       /* owner = the proof object created with this source */
void vmn_jrs_release_owner(void* owner);               /* called by the owner's _free wrapper (no-op for verifiers): random
                                                          source AND communicator bridges of that object */

/* vmn_comm over a Java object implementing com.verificatum.vmnhip.CommBridge:
 *     int rank(); int world(); byte[] allGather(byte[] mine)     (world * mine.length bytes, in rank order)
 * owned by the proof object it was set on (vmn_*_set_comm) and released with it. */
typedef struct vmn_jcomm vmn_jcomm;
vmn_jcomm* vmn_jcomm_new(JNIEnv* env, jobject bridge, void* owner);
void vmn_jcomm_fill(vmn_jcomm* h, vmn_comm* out);
#endif
