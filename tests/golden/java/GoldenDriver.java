// GoldenDriver -- OPT-IN harness, not part of the product and never run in the build image (no JDK there).
//
// Compiled TOGETHER WITH the reference's own three pure-JDK files (NetNode.java, NetMakerOriginal.java,
// NeighborNetCanonical.java, taken from $FASTNN_REF_DIR by make_java_golden.sh - nothing of the reference is
// copied into this repository) it runs the reference's Canonical path,
//     new NeighborNetCanonical(D, n, threads, pool).runNeighborNet()        (NetMakerOriginal.java:129)
// on the synthetic matrices of SURVEY.md 8(d) and prints one JSON line per case: the order's sha256 (over the
// n + 1 little-endian int32 values, as tests/golden/*.json hash it), the order itself for small n, and the
// seconds of the runNeighborNet() span (what FastNN.java:377-382 times).
// With `hip` as the first argument the SAME matrices also go through jni/NeighborNetHIP.java -> jni/fastnn_jni.c ->
// libfastnn_hip.so (the drop-in a maintainer would add), and the line carries both hashes and both timings.
// usage: java -Xmx<..>g nnet.GoldenDriver [hip] <threads> <n:dist:seed> [<n:dist:seed> ...]
package nnet;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.security.MessageDigest;
import java.util.concurrent.ExecutorService;
import java.util.concurrent.Executors;

public class GoldenDriver {
    static long state;

    static long next() {  // SplitMix64
        long z = (state += 0x9E3779B97F4A7C15L);
        z = (z ^ (z >>> 30)) * 0xBF58476D1CE4E5B9L;
        z = (z ^ (z >>> 27)) * 0x94D049BB133111EBL;
        return z ^ (z >>> 31);
    }

    static double[][] synth(int n, long seed, String dist) {
        state = seed;
        double[][] D = new double[n][n];
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) {
                double u = (double) (next() >>> 11) * 0x1.0p-53;
                double d = dist.equals("dec4") ? (double) ((long) (u * 1e4) + 1) / 1e4 : u + 0x1.0p-10;
                D[i][j] = d;
                D[j][i] = d;
            }
        return D;
    }

    static String sha(int[] order) throws Exception {
        ByteBuffer bb = ByteBuffer.allocate(4 * order.length).order(ByteOrder.LITTLE_ENDIAN);
        for (int v : order) bb.putInt(v);
        StringBuilder hex = new StringBuilder();
        for (byte b : MessageDigest.getInstance("SHA-256").digest(bb.array())) hex.append(String.format("%02x", b));
        return hex.toString();
    }

    public static void main(String[] a) throws Exception {
        boolean hip = a.length > 0 && a[0].equals("hip");
        if (hip) a = java.util.Arrays.copyOfRange(a, 1, a.length);
        int threads = Integer.parseInt(a[0]);
        ExecutorService pool = threads > 1 ? Executors.newFixedThreadPool(threads) : null;
        for (int c = 1; c < a.length; c++) {
            String[] p = a[c].split(":");
            int n = Integer.parseInt(p[0]);
            long seed = Long.parseLong(p[2]);
            double[][] D = synth(n, seed, p[1]);
            String hipPart = "";
            if (hip) {  // first: the reference destroys D in place (NetMakerOriginal.java:653-656), the engine does not
                long h0 = System.nanoTime();
                int[] oh = new NeighborNetHIP(D, n, threads, pool).runNeighborNet();
                hipPart = ",\"hip_order_sha256\":\"" + sha(oh) + "\",\"hip_seconds\":" + (System.nanoTime() - h0) * 1e-9;
            }
            long t0 = System.nanoTime();
            int[] order = new NeighborNetCanonical(D, n, threads, pool).runNeighborNet();
            double sec = (System.nanoTime() - t0) * 1e-9;
            String hex = sha(order);
            StringBuilder o = new StringBuilder("null");
            if (n <= 64) {
                o = new StringBuilder("[");
                for (int i = 0; i < order.length; i++) o.append(i > 0 ? "," : "").append(order[i]);
                o.append("]");
            }
            System.out.println("{\"n\":" + n + ",\"dist\":\"" + p[1] + "\",\"seed\":" + seed + ",\"threads\":" + threads
                    + ",\"order_sha256\":\"" + hex + "\",\"order\":" + o + ",\"seconds\":" + sec + hipPart + "}");
            System.out.flush();
        }
        System.exit(0);  // the reference never shuts its thread pool down (FastNN.java:536-538): leave explicitly
    }
}
