"""Second, independent CPU restatement of the reference's Canonical path (pure
Python objects and loops; small n only).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (no JVM in the build image, the
reference ships no golden vectors).  Written directly from the Java text,
independently of oracle/nnet_oracle.c, so that a transcription slip in either
shows up as a disagreement (tests/test_oracle.py).  Python floats are IEEE
binary64 and CPython never fuses a*b+c, which matches Java `double` arithmetic
(strictfp or not, SSE2 doubles round each operation to binary64).

Follows: NetNode.java:5-15, NetMakerOriginal.java:129-726,
NeighborNetCanonical.java:139-179 (numThreads == 1).
"""
from __future__ import annotations

import sys

DOUBLE_MAX = sys.float_info.max


class NetNode:  # NetNode.java:5-15
    __slots__ = ("id", "distID", "positionID", "nbr", "ch1", "ch2", "next", "prev", "Sx")

    def __init__(self):
        self.id = 0
        self.distID = 0
        self.positionID = 0
        self.nbr = None
        self.ch1 = None
        self.ch2 = None
        self.next = None
        self.prev = None
        self.Sx = 0.0


class NeighborNetCanonicalRef:
    """D: list of lists (or 2-D array) of floats, mutated in place like the Java."""

    def __init__(self, d, numTaxa):  # NetMakerOriginal.java:51-56
        self.ntax = numTaxa
        self.D = d
        self.best = 0.0
        self.Cx = None
        self.Cy = None
        self.trace = []  # (m, c, Cx.id, Cy.id, x.id, y.id, kind, u.id)

    # NetMakerOriginal.java:129-162
    def runNeighborNet(self):
        ntax = self.ntax
        if ntax <= 3:
            return [i for i in range(ntax + 1)]
        netNodes = [None] * ntax
        for i in range(ntax, 0, -1):
            taxNode = NetNode()
            taxNode.id = i
            taxNode.positionID = i - 1
            netNodes[i - 1] = taxNode
            taxNode.distID = i - 1
        amalgs = []
        num_nodes = ntax
        self.initialize(self.D, netNodes, num_nodes, num_nodes, num_nodes)
        num_nodes = self.agglomNodes(amalgs, self.D, netNodes, num_nodes)
        return self.expandNodes(num_nodes, ntax, amalgs, netNodes)

    # NetMakerOriginal.java:164-191
    def initialize(self, D, netNodes, num_nodes, num_active, num_clusters):
        for p in netNodes:
            if p.nbr is None or p.nbr.id > p.id:
                for j in range(p.positionID + 1, num_nodes):
                    q = netNodes[j]
                    if q.nbr is None or ((q.nbr.id > q.id) and (q.nbr is not p)):
                        if p.nbr is None and q.nbr is None:
                            Dpq = D[p.distID][q.distID]
                        elif p.nbr is not None and q.nbr is None:
                            Dpq = (D[p.distID][q.distID] + D[p.nbr.distID][q.distID]) / 2.0
                        elif p.nbr is None and q.nbr is not None:
                            Dpq = (D[p.distID][q.distID] + D[p.distID][q.nbr.distID]) / 2.0
                        else:
                            Dpq = (D[p.distID][q.distID] + D[p.distID][q.nbr.distID]
                                   + D[p.nbr.distID][q.distID] + D[p.nbr.distID][q.nbr.distID]) / 4.0
                        p.Sx += Dpq
                        if p.nbr is not None:
                            p.nbr.Sx += Dpq
                        q.Sx += Dpq
                        if q.nbr is not None:
                            q.nbr.Sx += Dpq

    # NeighborNetCanonical.java:139-179 (serial) == NetMakerOriginal.java:197-236
    def findNodes(self, D, netNodes, num_active, num_clusters):
        self.Cx = self.Cy = None
        self.best = DOUBLE_MAX
        for i in range(num_active):
            p = netNodes[i]
            if p.nbr is not None and p.nbr.id < p.id:
                continue
            for j in range(i):
                q = netNodes[j]
                if q.nbr is not None and q.nbr.id < q.id:
                    continue
                if q.nbr is p:
                    continue
                if p.nbr is None and q.nbr is None:
                    Dpq = D[p.distID][q.distID]
                elif p.nbr is not None and q.nbr is None:
                    Dpq = (D[p.distID][q.distID] + D[p.nbr.distID][q.distID]) / 2.0
                elif p.nbr is None and q.nbr is not None:
                    Dpq = (D[p.distID][q.distID] + D[p.distID][q.nbr.distID]) / 2.0
                else:
                    Dpq = (D[p.distID][q.distID] + D[p.distID][q.nbr.distID]
                           + D[p.nbr.distID][q.distID] + D[p.nbr.distID][q.nbr.distID]) / 4.0
                Qpq = (float(num_clusters) - 2.0) * Dpq - p.Sx - q.Sx
                if (self.Cx is None or Qpq < self.best) and (p.nbr is not q):
                    self.Cx = p
                    self.Cy = q
                    self.best = Qpq

    # NetMakerOriginal.java:246-325
    def expandNodes(self, num_nodes, ntax, amalgs, netNodes):
        ordering = [0] * (ntax + 1)
        x = netNodes[0]
        y = netNodes[1]
        z = netNodes[2]
        x.next = y
        y.next = z
        z.next = x
        x.prev = z
        y.prev = x
        z.prev = y
        while amalgs:
            u = amalgs.pop()
            v = u.nbr
            x = u.ch1
            y = u.ch2
            z = v.ch2
            if v is not u.next:
                u, v = v, u
                x, z = z, x
            x.prev = u.prev
            x.prev.next = x
            x.next = y
            y.prev = x
            y.next = z
            z.prev = y
            z.next = v.next
            z.next.prev = z
        while x.id != 1:
            x = x.next
        a = x
        t = 0
        while True:
            t += 1
            ordering[t] = a.id
            a = a.next
            if a is x:
                break
        return ordering

    # NetMakerOriginal.java:331-395
    def agglomNodes(self, amalgs, D, netNodes, num_nodes):
        num_active = num_nodes
        num_clusters = num_nodes
        while num_active > 3:
            if num_active == 4 and num_clusters == 2:
                p = netNodes[0]
                if p.nbr is not netNodes[1]:
                    q = netNodes[1]
                else:
                    q = netNodes[2]
                if (D[p.distID][q.distID] + D[p.nbr.distID][q.nbr.distID]
                        < D[p.distID][q.nbr.distID] + D[p.nbr.distID][q.distID]):
                    self.trace.append((num_active, num_clusters, 0, 0, p.id, q.id, 5, num_nodes + 1))
                    self.agg3way(p, q, q.nbr, amalgs, D, netNodes, num_nodes, num_active)
                    num_nodes += 2
                else:
                    self.trace.append((num_active, num_clusters, 0, 0, p.id, q.nbr.id, 5, num_nodes + 1))
                    self.agg3way(p, q.nbr, q, amalgs, D, netNodes, num_nodes, num_active)
                    num_nodes += 2
                break
            # num_active <= 1024 -> findNodesDefault, else findNodes: same scan (threads == 1)
            self.findNodes(D, netNodes, num_active, num_clusters)
            if self.Cx.id > self.Cy.id:
                self.Cx, self.Cy = self.Cy, self.Cx
            num_nodes, num_active, num_clusters = self.handleAgglomerationEvent(
                self.Cx, self.Cy, amalgs, D, netNodes, num_nodes, num_active, num_clusters)
        return num_nodes

    # NetMakerOriginal.java:397-515
    def handleAgglomerationEvent(self, Cx, Cy, amalgs, D, netNodes, num_nodes, num_active, num_clusters):
        m0, c0 = num_active, num_clusters
        x = Cx
        y = Cy
        Cx_Rx = 0.0
        Cx_nbr_Rx = 0.0
        Cy_Rx = 0.0
        Cy_nbr_Rx = 0.0
        if Cx.nbr is not None or Cy.nbr is not None:
            Cx_Rx = self.ComputeRx(Cx, Cx, Cy, D, netNodes, num_active)
            if Cx.nbr is not None:
                Cx_nbr_Rx = self.ComputeRx(Cx.nbr, Cx, Cy, D, netNodes, num_active)
            Cy_Rx = self.ComputeRx(Cy, Cx, Cy, D, netNodes, num_active)
            if Cy.nbr is not None:
                Cy_nbr_Rx = self.ComputeRx(Cy.nbr, Cx, Cy, D, netNodes, num_active)
        m = num_clusters
        if Cx.nbr is not None:
            m += 1
        if Cy.nbr is not None:
            m += 1
        self.best = (float(m) - 2.0) * D[Cx.distID][Cy.distID] - Cx_Rx - Cy_Rx
        if Cx.nbr is not None:
            Qpq = (float(m) - 2.0) * D[Cx.nbr.distID][Cy.distID] - Cx_nbr_Rx - Cy_Rx
            if Qpq < self.best:
                x = Cx.nbr
                y = Cy
                self.best = Qpq
        if Cy.nbr is not None:
            Qpq = (float(m) - 2.0) * D[Cx.distID][Cy.nbr.distID] - Cx_Rx - Cy_nbr_Rx
            if Qpq < self.best:
                x = Cx
                y = Cy.nbr
                self.best = Qpq
        if Cx.nbr is not None and Cy.nbr is not None:
            Qpq = (float(m) - 2.0) * D[Cx.nbr.distID][Cy.nbr.distID] - Cx_nbr_Rx - Cy_nbr_Rx
            if Qpq < self.best:
                x = Cx.nbr
                y = Cy.nbr
                self.best = Qpq
        for i in range(num_active):
            p = netNodes[i]
            if i != x.positionID and i != y.positionID:
                self.subtractClusterDistance(p, x)
                self.subtractClusterDistance(p, y)
        xid, yid = x.id, y.id
        if x.nbr is None and y.nbr is None:
            u = self.agg2way(x, y)
            num_clusters -= 1
            kind = 2
        elif x.nbr is None:
            u = self.agg3way(x, y, y.nbr, amalgs, D, netNodes, num_nodes, num_active)
            num_nodes += 2
            num_active -= 1
            num_clusters -= 1
            x.positionID = -1
            y.positionID = -1
            y.nbr.positionID = -1
            kind = 3
        elif y.nbr is None or num_active == 4:
            u = self.agg3way(y, x, x.nbr, amalgs, D, netNodes, num_nodes, num_active)
            num_nodes += 2
            num_active -= 1
            num_clusters -= 1
            x.positionID = -1
            y.positionID = -1
            x.nbr.positionID = -1
            kind = 3
        else:
            u = self.agg4way(x.nbr, x, y, y.nbr, amalgs, D, netNodes, num_nodes, num_active)
            num_nodes += 4
            num_active -= 2
            num_clusters -= 1
            kind = 4
        self.updateClusterDistances(u, D, netNodes, num_active)
        self.trace.append((m0, c0, Cx.id, Cy.id, xid, yid, kind, u.id))
        return num_nodes, num_active, num_clusters

    # NetMakerOriginal.java:517-536
    def updateClusterDistances(self, u, D, netNodes, num_active):
        u.Sx = 0.0
        u.nbr.Sx = 0.0
        for i in range(num_active):
            p = netNodes[i]
            if (p.nbr is None or p.nbr.id > p.id) and (u.nbr is not p) and (u is not p):
                if p.nbr is None:
                    Dpu = (D[p.distID][u.distID] + D[p.distID][u.nbr.distID]) / 2.0
                else:
                    Dpu = (D[p.distID][u.distID] + D[p.distID][u.nbr.distID]
                           + D[p.nbr.distID][u.distID] + D[p.nbr.distID][u.nbr.distID]) / 4.0
                p.Sx += Dpu
                if p.nbr is not None:
                    p.nbr.Sx += Dpu
                u.Sx += Dpu
        u.nbr.Sx = u.Sx

    # NetMakerOriginal.java:549-561
    def ComputeRx(self, z, Cx, Cy, D, netNodes, num_active):
        Rx = 0.0
        for i in range(num_active):
            p = netNodes[i]
            if p is Cx or p is Cx.nbr or p is Cy or p is Cy.nbr or p.nbr is None:
                Rx += D[z.distID][p.distID]
            else:
                Rx += D[z.distID][p.distID] / 2.0
        return Rx

    # NetMakerOriginal.java:570-577
    def agg2way(self, x, y):
        x.nbr = y
        y.nbr = x
        return x

    # NetMakerOriginal.java:589-674
    def agg3way(self, x, y, z, amalgs, D, netNodes, num_nodes, num_active):
        u = NetNode()
        u.id = num_nodes + 1
        u.ch1 = x
        u.ch2 = y
        v = NetNode()
        v.id = num_nodes + 2
        v.ch1 = y
        v.ch2 = z
        netNodes[x.positionID] = u
        u.positionID = x.positionID
        u.distID = x.distID
        netNodes[z.positionID] = v
        v.positionID = z.positionID
        v.distID = z.distID
        netNodes[y.positionID] = netNodes[num_active - 1]
        netNodes[y.positionID].positionID = y.positionID
        netNodes[num_active - 1] = None
        u.nbr = v
        v.nbr = u
        for i in range(num_active - 1):
            p = netNodes[i]
            val = (2.0 / 3.0) * D[x.distID][p.distID] + D[y.distID][p.distID] / 3.0
            D[p.distID][u.distID] = val
            D[u.distID][p.distID] = val
            val = (2.0 / 3.0) * D[z.distID][p.distID] + D[y.distID][p.distID] / 3.0
            D[p.distID][v.distID] = val
            D[v.distID][p.distID] = val
        D[v.distID][v.distID] = 0.0
        D[u.distID][u.distID] = 0.0
        amalgs.append(u)
        return u

    # NetMakerOriginal.java:681-696
    def subtractClusterDistance(self, p, x):
        D = self.D
        if p is not x and p is not x.nbr and (p.nbr is None or p.nbr.id > p.id):
            if p.nbr is None and x.nbr is None:
                Dpx = D[p.distID][x.distID]
            elif p.nbr is not None and x.nbr is None:
                Dpx = (D[p.distID][x.distID] + D[p.nbr.distID][x.distID]) / 2.0
            elif p.nbr is None and x.nbr is not None:
                Dpx = (D[p.distID][x.distID] + D[p.distID][x.nbr.distID]) / 2.0
            else:
                Dpx = (D[p.distID][x.distID] + D[p.distID][x.nbr.distID]
                       + D[p.nbr.distID][x.distID] + D[p.nbr.distID][x.nbr.distID]) / 4.0
            p.Sx -= Dpx
            if p.nbr is not None:
                p.nbr.Sx -= Dpx

    # NetMakerOriginal.java:707-726
    def agg4way(self, x2, x, y, y2, amalgs, D, netNodes, num_nodes, num_active):
        u = self.agg3way(x2, x, y, amalgs, D, netNodes, num_nodes, num_active)
        num_nodes += 2
        v = self.agg3way(u, u.nbr, y2, amalgs, D, netNodes, num_nodes, num_active - 1)
        num_nodes += 2
        x2.positionID = -1
        x.positionID = -1
        y.positionID = -1
        y2.positionID = -1
        u.positionID = -1
        u.nbr.positionID = -1
        return v


def run(D):
    """D: 2-D numpy array or list of lists. Returns (ordering list, trace list)."""
    d = [[float(v) for v in row] for row in D]
    eng = NeighborNetCanonicalRef(d, len(d))
    order = eng.runNeighborNet()
    return order, eng.trace
