// NeighborNetHIP -- the Java side of the drop-in: `-mode Canonical` (and `-mode Relaxed` without -additive) of
// JacobPorter/FastNeighborNet on an MI355X through libfastnn_hip.so (C ABI: include/fastnn.h; JNI shim: fastnn_jni.c).
//
// A maintainer of the reference adds this file to package `nnet` and changes ONE line of FastNN.main's mode switch
// (FastNN.java:324-328):   myNMO = new NeighborNetHIP(D, nTaxa, nThreads, pool);
// Everything else of the Java code base stays: the class plugs into the reference's own seam, the abstract class
// NetMakerOriginal (ctor NetMakerOriginal.java:51-56, runNeighborNet() :129-162, abstract findNodes :193).
//
// Not compiled in the build image of this repository (no JDK there); tests/golden/java/make_java_golden.sh compiles it
// together with the reference's sources and fastnn_jni.c wherever a JDK and FASTNN_REF_DIR exist, and
// GoldenDriver's "hip" engine then holds it to the reference's own order on the same matrices.
package nnet;

import java.util.Stack;
import java.util.concurrent.ExecutorService;

public class NeighborNetHIP extends NetMakerOriginal {
    static { System.loadLibrary("fastnn_jni"); }   // fastnn_jni.c; it links libfastnn_hip.so

    /** fnn_opts.mode: 0 = Canonical (NeighborNetCanonical.java), 1 = Relaxed without -additive (NeighborNetLocal.java:170-264). */
    private final int mode;
    /** Relaxed mode: seed of java.util.Random (the reference's ThreadLocalRandom, NeighborNetLocal.java:30, cannot be seeded). */
    private final long seed;
    private final int device;

    public NeighborNetHIP(double[][] d, int numTaxa, int numThreads, ExecutorService pool) {
        this(d, numTaxa, numThreads, pool, 0, 0L, 0);
    }

    public NeighborNetHIP(double[][] d, int numTaxa, int numThreads, ExecutorService pool, int mode, long seed, int device) {
        super(d, numTaxa, numThreads, pool);   // numThreads / pool are accepted and ignored: the loop runs on the device
        this.mode = mode;
        this.seed = seed;
        this.device = device;
    }

    // ---- the C ABI, one native method per entry point (include/fastnn.h) ----
    private static native long  create(int n, int device, int mode, long seed);          // fnn_create
    private static native void  setRow(long handle, int row, double[] values);           // fnn_set_rows (one row)
    private static native void  setPackedUpper(long handle, double[] distances);         // fnn_set_packed_upper
    private static native int[] run(long handle, int n);                                 // fnn_run
    private static native void  destroy(long handle);                                    // fnn_destroy
    /** fnn_split_weights_f64: x of FastNN.java:452 (live index order, :405-419) for the flat row-major n x n distances. */
    public  static native void  splitWeights(double[] flatD, int n, int[] ordering, int device, double[] x);

    /** NetMakerOriginal.java:129-162.  The caller's D is NOT destroyed (the reference overwrites it, :653-656). */
    @Override
    public int[] runNeighborNet() {
        if (ntax <= 3) {   // :133-140
            int[] o = new int[ntax + 1];
            for (int i = 0; i <= ntax; i++) o[i] = i;
            return o;
        }
        long h = create(ntax, device, mode, seed);
        try {
            for (int i = 0; i < ntax; i++) setRow(h, i, D[i]);   // 256 KiB per call at 32768 taxa (SURVEY.md H7)
            return run(h, ntax);
        } finally {
            destroy(h);
        }
    }

    /** The same from the reader's own container: DistancesAndNames.distances (packed strict upper triangle, :24-38). */
    public static int[] orderFromPackedUpper(double[] distances, int ntax, int device) {
        if (ntax <= 3) {
            int[] o = new int[ntax + 1];
            for (int i = 0; i <= ntax; i++) o[i] = i;
            return o;
        }
        long h = create(ntax, device, 0, 0L);
        try {
            setPackedUpper(h, distances);
            return run(h, ntax);
        } finally {
            destroy(h);
        }
    }

    @Override
    protected void findNodes(Stack<NetNode> amalgs, double[][] D, NetNode[] netNodes, int num_nodes, int num_active, int num_clusters) {
        throw new UnsupportedOperationException("the whole agglomeration loop runs on the device");
    }
}
