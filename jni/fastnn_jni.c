/* fastnn_jni.c -- logic-free JNI shim between nnet.NeighborNetHIP (NeighborNetHIP.java) and the C ABI of
 * libfastnn_hip.so (include/fastnn.h).  One function per native method; a negative fnn_status becomes a
 * RuntimeException carrying fnn_last_error().
 *
 *   gcc -shared -fPIC -O2 -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -Iinclude jni/fastnn_jni.c \
 *       -Lfastneighbornet_amd -lfastnn_hip -Wl,-rpath,'$ORIGIN' -o fastneighbornet_amd/libfastnn_jni.so
 *
 * Not compiled in the build image (no jni.h there); tests/golden/java/make_java_golden.sh builds it where a JDK exists. */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include "fastnn.h"

static void throw_rt(JNIEnv* e) {
    jclass c = (*e)->FindClass(e, "java/lang/RuntimeException");
    if (c) (*e)->ThrowNew(e, c, fnn_last_error());
}

JNIEXPORT jlong JNICALL Java_nnet_NeighborNetHIP_create(JNIEnv* e, jclass c, jint n, jint device, jint mode, jlong seed) {
    fnn_opts o;
    memset(&o, 0, sizeof(o));
    o.device = device;
    o.validate = 1;              /* finite, bit-symmetric, zero diagonal: what FastNN.java:307-312 always builds */
    o.mode = mode;               /* FNN_MODE_CANONICAL / FNN_MODE_RELAXED */
    o.relaxed_seed_lo = (uint32_t)((uint64_t)seed & 0xFFFFFFFFu);
    o.relaxed_seed_hi = (uint32_t)((uint64_t)seed >> 32);
    fnn_handle* h = 0;
    if (fnn_create(n, &o, &h) != FNN_OK) { throw_rt(e); return 0; }
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_nnet_NeighborNetHIP_setRow(JNIEnv* e, jclass c, jlong h, jint row, jdoubleArray v) {
    jint n = (*e)->GetArrayLength(e, v);
    jdouble* p = (*e)->GetPrimitiveArrayCritical(e, v, 0);
    if (!p) return;              /* OutOfMemoryError is pending */
    int rc = fnn_set_rows((fnn_handle*)(intptr_t)h, row, 1, p, n);
    (*e)->ReleasePrimitiveArrayCritical(e, v, p, JNI_ABORT);
    if (rc != FNN_OK) throw_rt(e);
}

/* DistancesAndNames.distances (packed strict upper triangle, DistancesAndNames.java:24-38) in one call */
JNIEXPORT void JNICALL Java_nnet_NeighborNetHIP_setPackedUpper(JNIEnv* e, jclass c, jlong h, jdoubleArray distances) {
    jdouble* p = (*e)->GetPrimitiveArrayCritical(e, distances, 0);
    if (!p) return;
    int rc = fnn_set_packed_upper((fnn_handle*)(intptr_t)h, p);
    (*e)->ReleasePrimitiveArrayCritical(e, distances, p, JNI_ABORT);
    if (rc != FNN_OK) throw_rt(e);
}

JNIEXPORT jintArray JNICALL Java_nnet_NeighborNetHIP_run(JNIEnv* e, jclass c, jlong h, jint n) {
    jintArray out = (*e)->NewIntArray(e, n + 1);
    if (!out) return 0;
    jint* p = (*e)->GetIntArrayElements(e, out, 0);
    if (!p) return 0;
    int rc = fnn_run((fnn_handle*)(intptr_t)h, (int32_t*)p, 0);
    (*e)->ReleaseIntArrayElements(e, out, p, 0);
    if (rc != FNN_OK) { throw_rt(e); return 0; }
    return out;
}

JNIEXPORT void JNICALL Java_nnet_NeighborNetHIP_destroy(JNIEnv* e, jclass c, jlong h) {
    if (h) fnn_destroy((fnn_handle*)(intptr_t)h);
}

/* FastNN.java:404-453 (dense design matrix + mySolver.solve()) -> x[] in the live index order */
JNIEXPORT void JNICALL Java_nnet_NeighborNetHIP_splitWeights(JNIEnv* e, jclass c, jdoubleArray flatD, jint n, jintArray ordering,
                                                             jint device, jdoubleArray x) {
    jdouble* d = (*e)->GetDoubleArrayElements(e, flatD, 0);
    jint* o = (*e)->GetIntArrayElements(e, ordering, 0);
    jdouble* w = (*e)->GetDoubleArrayElements(e, x, 0);
    int rc = FNN_ENOMEM;
    if (d && o && w) rc = fnn_split_weights_f64(d, n, n, (const int32_t*)o, device, w, 0);
    if (w) (*e)->ReleaseDoubleArrayElements(e, x, w, 0);
    if (o) (*e)->ReleaseIntArrayElements(e, ordering, o, JNI_ABORT);
    if (d) (*e)->ReleaseDoubleArrayElements(e, flatD, d, JNI_ABORT);
    if (rc != FNN_OK) throw_rt(e);
}
