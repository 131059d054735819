"""Synthetic (A, M) generator (numpy, CPU): the deterministic inputs of tests, ``bench.py`` and the examples.

Input generation only: no solver arithmetic.  Neither the oracle (``oracle/``, the CPU checker) nor the solver package
``lsa-fw_amd/Solver`` contains it; both are fed from here.

What it restates
----------------
The reference assembles the linearised Navier-Stokes pair ``(A, M)`` with dolfinx/UFL, which is absent
here, so the *matrix structure* is restated with numpy on a structured triangulation:

* forms and signs: ``/root/reference/FEM/operators.py:236-284`` (``VariationalForms``) combined as in
  ``FEM/operators.py:461-471`` (A) and ``:502`` (M):

      A = -(U . grad u', v) - (u' . grad U, v) - (1/Re)(grad u', grad v) + (p', div v) + (q, div u')
      M = (u', v)

* spaces: Taylor-Hood P2 vector velocity + P1 pressure in one mixed space
  (``FEM/spaces.py:112-124,175-177``); the mixed-cell sparsity pattern is the full 15x15 coupling per
  triangle, so the pressure-pressure block and the cross-component mass entries are *stored explicit
  zeros* (``tests/unit/FEM/test_operators.py:179-182,209-210``) and A and M share one pattern;
* Dirichlet dofs: ``assemble_matrix(form, bcs=...)`` zeroes the row and column and puts 1.0 on the
  diagonal of A **and** M (``FEM/operators.py:483-485,504-506``), hence the spurious lambda = 1 modes
  noted in ``tests/benchmark/vibrating_membrane.py:169-173``;
* scalar P2 Laplace / mass pair of the membrane benchmark (``tests/benchmark/vibrating_membrane.py:103-125``).

Dof order is dolfinx-like node-interleaved: node n carries ``[ux, uy]`` (+ ``p`` on mesh vertices).
No RNG is used anywhere: the matrices are a pure function of the arguments.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

# --------------------------------------------------------------------------------------------------
# reference-element tables
# --------------------------------------------------------------------------------------------------


def _duffy_rule(n: int = 4) -> tuple[np.ndarray, np.ndarray]:
    """Collapsed Gauss-Legendre rule on the unit triangle (exact to degree 2n-2). Weights sum to 1/2."""
    g, w = np.polynomial.legendre.leggauss(n)
    g = 0.5 * (g + 1.0)
    w = 0.5 * w
    u, v = np.meshgrid(g, g, indexing="ij")
    wu, wv = np.meshgrid(w, w, indexing="ij")
    l1 = u.ravel()
    l2 = (v * (1.0 - u)).ravel()
    wt = (wu * wv * (1.0 - u)).ravel()
    lam = np.stack([1.0 - l1 - l2, l1, l2], axis=1)  # (Q, 3) barycentric
    return lam, wt


def _p2_tables(lam: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """P2 basis (Q,6) and d(basis)/d(lambda_i) (Q,6,3). Order: v0 v1 v2 e01 e12 e20."""
    l0, l1, l2 = lam[:, 0], lam[:, 1], lam[:, 2]
    phi = np.stack(
        [l0 * (2 * l0 - 1), l1 * (2 * l1 - 1), l2 * (2 * l2 - 1), 4 * l0 * l1, 4 * l1 * l2, 4 * l2 * l0],
        axis=1,
    )
    z = np.zeros_like(l0)
    d = np.zeros((lam.shape[0], 6, 3))
    d[:, 0] = np.stack([4 * l0 - 1, z, z], axis=1)
    d[:, 1] = np.stack([z, 4 * l1 - 1, z], axis=1)
    d[:, 2] = np.stack([z, z, 4 * l2 - 1], axis=1)
    d[:, 3] = np.stack([4 * l1, 4 * l0, z], axis=1)
    d[:, 4] = np.stack([z, 4 * l2, 4 * l1], axis=1)
    d[:, 5] = np.stack([4 * l2, z, 4 * l0], axis=1)
    return phi, d


# --------------------------------------------------------------------------------------------------
# structured mesh
# --------------------------------------------------------------------------------------------------


@dataclass
class StructuredMesh:
    """nx x ny cells, each split into two triangles; P2 nodes live on a (2nx+1) x (2ny+1) lattice."""

    nx: int
    ny: int
    xs: np.ndarray  # (2nx+1,) lattice abscissae (odd entries are midpoints)
    ys: np.ndarray  # (2ny+1,)

    @property
    def n_nodes(self) -> int:
        return (2 * self.nx + 1) * (2 * self.ny + 1)

    def node(self, i, j):
        """Lattice (i, j) -> node id; y runs fastest (ny <= nx keeps the bandwidth small)."""
        return i * (2 * self.ny + 1) + j

    def node_xy(self) -> tuple[np.ndarray, np.ndarray]:
        X, Y = np.meshgrid(self.xs, self.ys, indexing="ij")
        return X.ravel(), Y.ravel()

    def is_vertex(self) -> np.ndarray:
        I, J = np.meshgrid(np.arange(2 * self.nx + 1), np.arange(2 * self.ny + 1), indexing="ij")
        return ((I % 2 == 0) & (J % 2 == 0)).ravel()

    def triangles(self) -> np.ndarray:
        """(E, 6) node ids per triangle in P2 order v0 v1 v2 e01 e12 e20 (counter-clockwise)."""
        ci, cj = np.meshgrid(np.arange(self.nx), np.arange(self.ny), indexing="ij")
        i0, j0 = 2 * ci.ravel(), 2 * cj.ravel()
        n = self.node
        # lower-right triangle (00, 20, 22) and upper-left triangle (00, 22, 02)
        t1 = np.stack(
            [n(i0, j0), n(i0 + 2, j0), n(i0 + 2, j0 + 2), n(i0 + 1, j0), n(i0 + 2, j0 + 1), n(i0 + 1, j0 + 1)], axis=1
        )
        t2 = np.stack(
            [n(i0, j0), n(i0 + 2, j0 + 2), n(i0, j0 + 2), n(i0 + 1, j0 + 1), n(i0 + 1, j0 + 2), n(i0, j0 + 1)], axis=1
        )
        return np.concatenate([t1, t2], axis=0)


def _lattice(edges: np.ndarray) -> np.ndarray:
    out = np.empty(2 * len(edges) - 1)
    out[0::2] = edges
    out[1::2] = 0.5 * (edges[:-1] + edges[1:])
    return out


def graded_edges(lo: float, hi: float, n: int, centre: float, strength: float) -> np.ndarray:
    """n+1 cell edges on [lo, hi], sinh-clustered around ``centre`` (strength 0 = uniform)."""
    if strength <= 0.0:
        return np.linspace(lo, hi, n + 1)
    a = np.arcsinh((lo - centre) * strength)
    b = np.arcsinh((hi - centre) * strength)
    t = np.linspace(a, b, n + 1)
    e = centre + np.sinh(t) / strength
    e[0], e[-1] = lo, hi
    return e


def channel_mesh(nx: int, ny: int, *, x_range=(-40.0, 120.0), y_range=(-40.0, 40.0), grading: float = 1.0):
    """Cylinder-channel box of ``config_files/2D/cylinder/geometry.toml:4-5`` graded towards (0, 0)."""
    ex = graded_edges(x_range[0], x_range[1], nx, 0.0, grading)
    ey = graded_edges(y_range[0], y_range[1], ny, 0.0, grading)
    return StructuredMesh(nx, ny, _lattice(ex), _lattice(ey))


# --------------------------------------------------------------------------------------------------
# element kernels (vectorised over elements)
# --------------------------------------------------------------------------------------------------


def _geometry(mesh: StructuredMesh, tri: np.ndarray):
    X, Y = mesh.node_xy()
    x = X[tri[:, :3]]
    y = Y[tri[:, :3]]
    j11 = x[:, 1] - x[:, 0]
    j12 = x[:, 2] - x[:, 0]
    j21 = y[:, 1] - y[:, 0]
    j22 = y[:, 2] - y[:, 0]
    det = j11 * j22 - j12 * j21
    if np.any(det <= 0):
        raise ValueError("degenerate or clockwise triangle")
    # gradients of barycentric coordinates, (E, 3, 2)
    gl = np.empty((len(tri), 3, 2))
    gl[:, 1, 0] = j22 / det
    gl[:, 1, 1] = -j12 / det
    gl[:, 2, 0] = -j21 / det
    gl[:, 2, 1] = j11 / det
    gl[:, 0] = -gl[:, 1] - gl[:, 2]
    return det, gl


@dataclass
class EigenSystem:
    """Assembled pair plus the index sets callers of the reference path use."""

    A: sp.csr_matrix
    M: sp.csr_matrix
    dofs_u: np.ndarray
    dofs_p: np.ndarray
    dirichlet: np.ndarray
    mesh: StructuredMesh
    node_offset: np.ndarray  # first dof of each node

    @property
    def n(self) -> int:
        return self.A.shape[0]


def wake_baseflow(X: np.ndarray, Y: np.ndarray, *, deficit: float = 1.25, width: float = 0.9, decay: float = 12.0):
    """Analytic cylinder-wake-like base flow U = (1 - a(x) exp(-y^2 / 2 s^2), 0), a = 0 upstream."""
    a = deficit * np.exp(-np.maximum(X, 0.0) / decay) * (X > -0.5)
    ux = 1.0 - a * np.exp(-(Y**2) / (2.0 * width**2))
    return ux, np.zeros_like(ux)


def assemble_linearized_ns(
    mesh: StructuredMesh,
    re: float = 50.0,
    *,
    cylinder_radius: float = 0.5,
    baseflow=wake_baseflow,
    chunk: int = 200_000,
) -> EigenSystem:
    """Assemble (A, M) of the linearised Navier-Stokes operator; see the module docstring for the forms."""
    tri_all = mesh.triangles()
    X, Y = mesh.node_xy()
    isv = mesh.is_vertex()
    n_nodes = mesh.n_nodes
    node_offset = 2 * np.arange(n_nodes) + np.concatenate([[0], np.cumsum(isv)[:-1]])
    n = int(2 * n_nodes + isv.sum())

    Ux, Uy = baseflow(X, Y)
    inside = X**2 + Y**2 <= cylinder_radius**2
    Ux = np.where(inside, 0.0, Ux)
    Uy = np.where(inside, 0.0, Uy)

    lam, wq = _duffy_rule(4)
    phi, dphi = _p2_tables(lam)  # (Q,6), (Q,6,3)
    psi = lam  # P1 basis (Q,3)

    rows_l, cols_l, a_l, m_l = [], [], [], []
    for s in range(0, len(tri_all), chunk):
        tri = tri_all[s : s + chunk]
        E = len(tri)
        det, gl = _geometry(mesh, tri)
        w = wq[None, :] * det[:, None]  # (E,Q)  (sum wq = 1/2 -> triangle area)
        g = np.einsum("qai,eik->eqak", dphi, gl)  # physical gradients of P2 basis (E,Q,6,2)
        ux_q = np.einsum("qa,ea->eq", phi, Ux[tri])
        uy_q = np.einsum("qa,ea->eq", phi, Uy[tri])
        dux = np.einsum("eqak,ea->eqk", g, Ux[tri])  # grad Ux (E,Q,2)
        duy = np.einsum("eqak,ea->eqk", g, Uy[tri])

        mass = np.einsum("eq,qa,qb->eab", w, phi, phi)
        stiff = np.einsum("eq,eqak,eqbk->eab", w, g, g)
        adv = ux_q[:, :, None] * g[..., 0] + uy_q[:, :, None] * g[..., 1]  # U.grad(phi_b)  (E,Q,6)
        conv = np.einsum("eq,qa,eqb->eab", w, phi, adv)
        # shear[i][j]_ab = (phi_b dU_i/dx_j, phi_a)
        sh = [[np.einsum("eq,eq,qa,qb->eab", w, d[..., j], phi, phi) for j in range(2)] for d in (dux, duy)]
        # G[i]_ac = (psi_c, d phi_a / dx_i)
        G = [np.einsum("eq,qc,eqa->eac", w, psi, g[..., i]) for i in range(2)]

        Ae = np.zeros((E, 15, 15))
        Me = np.zeros((E, 15, 15))
        # local layout: [ux(6), uy(6), p(3)]
        for i in range(2):
            si = slice(6 * i, 6 * i + 6)
            for j in range(2):
                sj = slice(6 * j, 6 * j + 6)
                Ae[:, si, sj] -= sh[i][j]
            Ae[:, si, si] -= conv + stiff / re
            Me[:, si, si] = mass
            Ae[:, si, 12:15] += G[i]
            Ae[:, 12:15, si] += np.transpose(G[i], (0, 2, 1))

        ldofs = np.concatenate(
            [node_offset[tri], node_offset[tri] + 1, node_offset[tri[:, :3]] + 2], axis=1
        )  # (E,15)
        rows_l.append(np.repeat(ldofs, 15, axis=1).ravel())
        cols_l.append(np.tile(ldofs, (1, 15)).ravel())
        a_l.append(Ae.ravel())
        m_l.append(Me.ravel())

    rows = np.concatenate(rows_l)
    cols = np.concatenate(cols_l)
    A = sp.coo_matrix((np.concatenate(a_l), (rows, cols)), shape=(n, n)).tocsr()
    M = sp.coo_matrix((np.concatenate(m_l), (rows, cols)), shape=(n, n)).tocsr()
    A.sort_indices()
    M.sort_indices()
    assert np.array_equal(A.indptr, M.indptr) and np.array_equal(A.indices, M.indices)

    # Dirichlet velocity: inlet (marker 1) + cylinder wall (marker 5), bcs_perturbation.toml:1-10
    dnodes = np.flatnonzero((np.isclose(X, mesh.xs[0])) | inside)
    ddofs = np.sort(np.concatenate([node_offset[dnodes], node_offset[dnodes] + 1]))
    # Pressure dofs of the masked ("solid") region.  A body-fitted mesh has no dofs inside the cylinder; on this
    # structured mesh the pressure dofs at masked vertices, and any other pressure dof whose every velocity neighbour
    # is pinned, would be left with an all-zero or nearly all-zero row (a handful of couplings shared by several such
    # dofs), which makes leading blocks of A - sigma M singular.  They are pinned like the velocity dofs.
    dofs_p = node_offset[isv] + 2
    dflag = np.zeros(n, dtype=bool)
    dflag[ddofs] = True
    live = np.add.reduceat((~dflag[A.indices]) & (A.data != 0.0), A.indptr[:-1]) > 0
    masked_vertex_p = node_offset[np.flatnonzero(inside & isv)] + 2
    dead_p = np.union1d(dofs_p[~live[dofs_p]], masked_vertex_p)
    ddofs = np.sort(np.concatenate([ddofs, dead_p]))
    A = _apply_dirichlet(A, ddofs)
    M = _apply_dirichlet(M, ddofs)

    mask = np.ones(n, dtype=bool)
    mask[dofs_p] = False
    dofs_u = np.flatnonzero(mask)
    return EigenSystem(A, M, dofs_u.astype(np.int32), dofs_p.astype(np.int32), ddofs.astype(np.int32), mesh, node_offset)


def _apply_dirichlet(A: sp.csr_matrix, dofs: np.ndarray) -> sp.csr_matrix:
    """Zero rows and columns of ``dofs`` (entries stay in the pattern) and put 1 on their diagonal."""
    n = A.shape[0]
    flag = np.zeros(n, dtype=bool)
    flag[dofs] = True
    row_of = np.repeat(np.arange(n), np.diff(A.indptr))
    data = A.data.copy()
    kill = flag[row_of] | flag[A.indices]
    data[kill] = 0.0
    data[kill & (row_of == A.indices)] = 1.0
    out = sp.csr_matrix((data, A.indices.copy(), A.indptr.copy()), shape=A.shape)
    return out


# --------------------------------------------------------------------------------------------------
# named synthetic cases
# --------------------------------------------------------------------------------------------------

CASES = {
    # name: (nx, ny)
    "S2k": (20, 10),
    "S5k": (32, 16),
    "S30k": (82, 41),
    "S120k": (164, 82),
    "S500k": (334, 167),
}

GRADING = 0.3  # one sinh-grading for the whole family, so the sizes are true refinements of each other

SIGMA_RE50 = 0.018 + 0.7379601143282424j  # /root/reference/.examples/eigenvalues.py:40 (Re = 50 target)


def cylinder_case(name: str = "S30k", re: float = 50.0) -> EigenSystem:
    nx, ny = CASES[name]
    return assemble_linearized_ns(channel_mesh(nx, ny, grading=GRADING), re)


def channel_pattern(nx: int, ny: int) -> sp.csr_matrix:
    """Sparsity pattern of :func:`assemble_linearized_ns` on the ``nx`` x ``ny`` channel mesh without the arithmetic (values all
    one): two dofs are coupled iff their nodes share a triangle (stored zeros included), same node-interleaved dof numbering.
    Seconds at 5 M unknowns, where the assembly takes minutes: the single-mesh form of the SpMV roofline matrix."""
    mesh = channel_mesh(nx, ny, grading=GRADING)
    tri = mesh.triangles().astype(np.int64)
    nn = mesh.n_nodes
    key = np.unique(np.concatenate([np.unique((tri[:, a][:, None] * nn + tri).ravel()) for a in range(6)]))
    na, nb = key // nn, key % nn
    isv = mesh.is_vertex()
    dpn = 2 + isv.astype(np.int64)  # dofs per node: (ux, uy[, p])
    node_offset = 2 * np.arange(nn, dtype=np.int64) + np.concatenate([[0], np.cumsum(isv)[:-1]])
    n = int(2 * nn + isv.sum())
    nptr = np.concatenate([[0], np.cumsum(np.bincount(na, minlength=nn))])
    width = dpn[nb]
    cptr = np.concatenate([[0], np.cumsum(width)])
    cols_node = np.repeat(node_offset[nb] - cptr[:-1], width) + np.arange(cptr[-1])
    row_len_node = cptr[nptr[1:]] - cptr[nptr[:-1]]
    rows_per_dof = np.repeat(row_len_node, dpn)
    indptr = np.concatenate([[0], np.cumsum(rows_per_dof)])
    start_node = np.repeat(cptr[nptr[:-1]], dpn)
    idx = np.repeat(start_node - indptr[:-1], rows_per_dof) + np.arange(indptr[-1])
    return sp.csr_matrix((np.ones(len(idx), dtype=np.int8), cols_node[idx].astype(np.int32), indptr.astype(np.int32)), shape=(n, n))


# --------------------------------------------------------------------------------------------------
# 3D: unit-cube duct, Taylor-Hood P2/P1 on Kuhn tetrahedra (BASELINE config 4)
# --------------------------------------------------------------------------------------------------
# The reference's 3D case is the unit cube cut into tetrahedra (``.examples/cube.py:37``: ``Mesher(Shape.UNIT_CUBE,
# (20, 20, 20), iCellType.TETRAHEDRON)``, Re = 10) with ``config_files/3D/unit_cube/mesh_tags.toml`` /``bcs.toml``:
# velocity Dirichlet on the inlet x = 0 and on the four walls, pressure Dirichlet on the outlet x = 1.  Restated here
# on n^3 cubes of six Kuhn tetrahedra each (all share the cube's main diagonal); the P2 nodes are exactly the
# (2n + 1)^3 lattice (cube-edge, face-diagonal and body-diagonal midpoints).  Row degrees: 3 x 10 + 4 = 34 unknowns per
# tetrahedron, mean row length ~ 95 at 5 M unknowns (SURVEY.md 8d quotes ~ 99).


def _tet_rule(n: int = 4) -> tuple[np.ndarray, np.ndarray]:
    """Collapsed Gauss-Legendre rule on the unit tetrahedron, exact to degree 2n - 3 (5 for n = 4: the convective and
    shear terms with a P2 base flow).  Returns barycentric points (Q, 4) and weights summing to 1/6."""
    g, w = np.polynomial.legendre.leggauss(n)
    g, w = 0.5 * (g + 1.0), 0.5 * w
    u, v, t = np.meshgrid(g, g, g, indexing="ij")
    wu, wv, wt = np.meshgrid(w, w, w, indexing="ij")
    x = u.ravel()
    y = (v * (1.0 - u)).ravel()
    z = (t * (1.0 - u) * (1.0 - v)).ravel()
    wq = (wu * wv * wt * (1.0 - u) ** 2 * (1.0 - v)).ravel()
    lam = np.stack([1.0 - x - y - z, x, y, z], axis=1)
    return lam, wq


_TET_EDGES = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))


def _p2_tet_tables(lam: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """P2 basis (Q, 10) and d(basis)/d(lambda_i) (Q, 10, 4) on a tetrahedron; order: 4 vertices, then the edges
    01 02 03 12 13 23."""
    Q = lam.shape[0]
    phi = np.empty((Q, 10))
    d = np.zeros((Q, 10, 4))
    for i in range(4):
        phi[:, i] = lam[:, i] * (2.0 * lam[:, i] - 1.0)
        d[:, i, i] = 4.0 * lam[:, i] - 1.0
    for e, (i, j) in enumerate(_TET_EDGES):
        phi[:, 4 + e] = 4.0 * lam[:, i] * lam[:, j]
        d[:, 4 + e, i] = 4.0 * lam[:, j]
        d[:, 4 + e, j] = 4.0 * lam[:, i]
    return phi, d


@dataclass
class CubeMesh:
    """n^3 cubes of the unit cube, six Kuhn tetrahedra each; P2 nodes on the (2n + 1)^3 lattice, z fastest."""

    n: int

    @property
    def n_nodes(self) -> int:
        return (2 * self.n + 1) ** 3

    def node(self, i, j, k):
        L = 2 * self.n + 1
        return (i * L + j) * L + k

    def node_xyz(self) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        g = np.linspace(0.0, 1.0, 2 * self.n + 1)
        X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
        return X.ravel(), Y.ravel(), Z.ravel()

    def is_vertex(self) -> np.ndarray:
        a = np.arange(2 * self.n + 1)
        I, J, K = np.meshgrid(a, a, a, indexing="ij")
        return ((I % 2 == 0) & (J % 2 == 0) & (K % 2 == 0)).ravel()

    def tetrahedra(self) -> np.ndarray:
        """(E, 10) node ids per tetrahedron in P2 order (positively oriented)."""
        import itertools

        a = np.arange(self.n)
        ci, cj, ck = (2 * g.ravel() for g in np.meshgrid(a, a, a, indexing="ij"))
        base = np.stack([ci, cj, ck], axis=1)  # lattice position of every cube's (0, 0, 0) corner
        out = []
        for perm in itertools.permutations(range(3)):
            # Kuhn path (0,0,0) -> +e_perm[0] -> +e_perm[1] -> (1,1,1); odd permutations are mirrored: swap two vertices
            corners = [np.zeros(3, dtype=np.int64)]
            for ax in perm:
                nxt = corners[-1].copy()
                nxt[ax] += 2
                corners.append(nxt)
            sign = np.linalg.det(np.array([c - corners[0] for c in corners[1:]], dtype=float))
            if sign < 0:
                corners[2], corners[3] = corners[3], corners[2]
            loc = corners + [(corners[i] + corners[j]) // 2 for i, j in _TET_EDGES]
            out.append(np.stack([self.node(base[:, 0] + l[0], base[:, 1] + l[1], base[:, 2] + l[2]) for l in loc], axis=1))
        return np.concatenate(out, axis=0)


def duct_baseflow(X, Y, Z, *, peak: float = 1.0):
    """Analytic fully developed-like duct profile U = (peak * 16 y (1 - y) z (1 - z), 0, 0): no slip on the four walls."""
    return peak * 16.0 * Y * (1.0 - Y) * Z * (1.0 - Z), np.zeros_like(X), np.zeros_like(X)


def assemble_linearized_ns_3d(mesh: CubeMesh, re: float = 10.0, *, baseflow=duct_baseflow, chunk: int = 20_000) -> EigenSystem:
    """(A, M) of the linearised Navier-Stokes operator on the unit cube: the forms of the module docstring in 3D, mixed
    34 x 34 coupling per tetrahedron (stored zeros in the pressure block and the cross-component mass entries)."""
    tets = mesh.tetrahedra()
    X, Y, Z = mesh.node_xyz()
    P = np.stack([X, Y, Z], axis=1)
    isv = mesh.is_vertex()
    n_nodes = mesh.n_nodes
    node_offset = 3 * np.arange(n_nodes) + np.concatenate([[0], np.cumsum(isv)[:-1]])
    n = int(3 * n_nodes + isv.sum())
    U = np.stack(baseflow(X, Y, Z), axis=1)  # (nodes, 3)

    lam, wq = _tet_rule(4)
    phi, dphi = _p2_tet_tables(lam)  # (Q,10), (Q,10,4)
    psi = lam  # P1 basis (Q,4)

    rows_l, cols_l, a_l, m_l = [], [], [], []
    for s0 in range(0, len(tets), chunk):
        tet = tets[s0 : s0 + chunk]
        E = len(tet)
        v = P[tet[:, :4]]  # (E,4,3)
        J = np.stack([v[:, 1] - v[:, 0], v[:, 2] - v[:, 0], v[:, 3] - v[:, 0]], axis=2)  # columns = edge vectors
        det = np.linalg.det(J)
        if np.any(det <= 0):
            raise ValueError("degenerate or inverted tetrahedron")
        Jinv = np.linalg.inv(J)  # rows = gradients of lambda_1..3
        gl = np.empty((E, 4, 3))
        gl[:, 1:] = Jinv
        gl[:, 0] = -Jinv.sum(axis=1)
        w = wq[None, :] * det[:, None]  # (E,Q): sum wq = 1/6 -> volume
        g = np.einsum("qai,eik->eqak", dphi, gl)  # physical gradients of the P2 basis (E,Q,10,3)
        Uq = np.einsum("qa,eac->eqc", phi, U[tet])  # (E,Q,3)
        dU = np.einsum("eqak,eac->eqck", g, U[tet])  # dU_c/dx_k (E,Q,3,3)
        mass = np.einsum("eq,qa,qb->eab", w, phi, phi)
        stiff = np.einsum("eq,eqak,eqbk->eab", w, g, g)
        adv = np.einsum("eqk,eqbk->eqb", Uq, g)  # U.grad(phi_b)
        conv = np.einsum("eq,qa,eqb->eab", w, phi, adv)
        Ae = np.zeros((E, 34, 34))
        Me = np.zeros((E, 34, 34))
        for i in range(3):
            si = slice(10 * i, 10 * i + 10)
            for j in range(3):
                sj = slice(10 * j, 10 * j + 10)
                Ae[:, si, sj] -= np.einsum("eq,eq,qa,qb->eab", w, dU[:, :, i, j], phi, phi)  # (phi_b dU_i/dx_j, phi_a)
            Ae[:, si, si] -= conv + stiff / re
            Me[:, si, si] = mass
            G = np.einsum("eq,qc,eqa->eac", w, psi, g[..., i])  # (psi_c, d phi_a / dx_i)
            Ae[:, si, 30:34] += G
            Ae[:, 30:34, si] += np.transpose(G, (0, 2, 1))
        ldofs = np.concatenate([node_offset[tet], node_offset[tet] + 1, node_offset[tet] + 2, node_offset[tet[:, :4]] + 3], axis=1)  # (E,34)
        rows_l.append(np.repeat(ldofs, 34, axis=1).ravel().astype(np.int32))
        cols_l.append(np.tile(ldofs, (1, 34)).ravel().astype(np.int32))
        a_l.append(Ae.ravel())
        m_l.append(Me.ravel())

    rows = np.concatenate(rows_l)
    cols = np.concatenate(cols_l)
    A = sp.coo_matrix((np.concatenate(a_l), (rows, cols)), shape=(n, n)).tocsr()
    M = sp.coo_matrix((np.concatenate(m_l), (rows, cols)), shape=(n, n)).tocsr()
    A.sort_indices()
    M.sort_indices()
    assert np.array_equal(A.indptr, M.indptr) and np.array_equal(A.indices, M.indices)

    # velocity Dirichlet: inlet x = 0 (marker 1) and the walls y, z in {0, 1} (markers 3-6); pressure Dirichlet on the
    # outlet x = 1 (marker 2): config_files/3D/unit_cube/bcs.toml, homogeneous for the perturbation
    tol = 1e-12
    vel_nodes = np.flatnonzero((X < tol) | (Y < tol) | (Y > 1 - tol) | (Z < tol) | (Z > 1 - tol))
    ddofs = np.concatenate([node_offset[vel_nodes] + c for c in range(3)])
    dofs_p = node_offset[isv] + 3
    out_p = node_offset[np.flatnonzero(isv & (X > 1 - tol))] + 3
    ddofs = np.sort(np.concatenate([ddofs, out_p]))
    dflag = np.zeros(n, dtype=bool)
    dflag[ddofs] = True
    live = np.add.reduceat((~dflag[A.indices]) & (A.data != 0.0), A.indptr[:-1]) > 0
    ddofs = np.union1d(ddofs, dofs_p[~live[dofs_p]])  # pressure unknowns with every velocity neighbour pinned
    A = _apply_dirichlet(A, ddofs)
    M = _apply_dirichlet(M, ddofs)
    mask = np.ones(n, dtype=bool)
    mask[dofs_p] = False
    return EigenSystem(A, M, np.flatnonzero(mask).astype(np.int32), dofs_p.astype(np.int32), ddofs.astype(np.int32), mesh, node_offset)


def cube_pattern(n_cells: int) -> sp.csr_matrix:
    """Sparsity pattern of :func:`assemble_linearized_ns_3d` on ``CubeMesh(n_cells)`` without the arithmetic (values all one):
    two dofs are coupled iff their nodes share a tetrahedron (stored zeros included), same dof numbering.  What the pattern-only
    analysis of the direct solver needs; minutes instead of hours at the size of BASELINE config 4 (58 cells: 4.9 M unknowns,
    4.8e8 entries)."""
    mesh = CubeMesh(n_cells)
    tets = mesh.tetrahedra().astype(np.int64)
    nn = mesh.n_nodes
    keys = []
    for a in range(10):  # node pairs sharing a tetrahedron, as sorted unique keys
        keys.append(np.unique((tets[:, a][:, None] * nn + tets).ravel()))
    key = np.unique(np.concatenate(keys))
    del keys
    na, nb = key // nn, key % nn
    isv = mesh.is_vertex()
    dpn = 3 + isv.astype(np.int64)  # dofs per node
    node_offset = 3 * np.arange(nn, dtype=np.int64) + np.concatenate([[0], np.cumsum(isv)[:-1]])
    n = int(3 * nn + isv.sum())
    # node-level CSR (neighbours sorted), then every neighbour b expands into its dofs, every row of node a repeats that list
    nptr = np.concatenate([[0], np.cumsum(np.bincount(na, minlength=nn))])
    width = dpn[nb]
    cptr = np.concatenate([[0], np.cumsum(width)])
    cols_node = np.repeat(node_offset[nb] - cptr[:-1], width) + np.arange(cptr[-1])  # dofs of all neighbours, node row by node row
    row_len_node = cptr[nptr[1:]] - cptr[nptr[:-1]]  # entries of one dof row of node a
    rows_per_dof = np.repeat(row_len_node, dpn)
    indptr = np.concatenate([[0], np.cumsum(rows_per_dof)])
    start_node = np.repeat(cptr[nptr[:-1]], dpn)  # where the node's column list starts, per dof row
    idx = np.repeat(start_node - indptr[:-1], rows_per_dof) + np.arange(indptr[-1])
    indices = cols_node[idx].astype(np.int32)
    return sp.csr_matrix((np.ones(len(indices), dtype=np.int8), indices, indptr.astype(np.int64 if indptr[-1] >= 2**31 else np.int32)), shape=(n, n))


CUBE_CASES = {"C2k": 4, "C9k": 7, "C20k": 9, "C40k": 11, "C80k": 14, "C160k": 18, "C300k": 22, "C640k": 29, "C1M": 34, "C2M": 43, "C5M": 58}
SIGMA_CUBE = -5.0  # shift of the 3D case: next to the least stable physical modes of the Re = 10 duct (-5.99, -6.00, -6.8, ...) and
# away from the spurious lambda = 1 of the identity Dirichlet rows; real, so the factors are float64


def cube_case(name: str = "C20k", re: float = 10.0) -> EigenSystem:
    return assemble_linearized_ns_3d(CubeMesh(CUBE_CASES[name]), re)


# --------------------------------------------------------------------------------------------------
# membrane benchmark (scalar P2 Laplace)
# --------------------------------------------------------------------------------------------------


def assemble_membrane(nx: int = 32, ny: int = 32, a: float = 2.0, b: float = 4.0):
    """(A, M) = (stiffness, mass) with identity Dirichlet rows on the whole boundary.

    Restates ``tests/benchmark/vibrating_membrane.py:103-125`` (forms ``FEM/operators.py:239-240,281-284``).
    """
    mesh = StructuredMesh(nx, ny, _lattice(np.linspace(0, a, nx + 1)), _lattice(np.linspace(0, b, ny + 1)))
    tri = mesh.triangles()
    lam, wq = _duffy_rule(4)
    phi, dphi = _p2_tables(lam)
    det, gl = _geometry(mesh, tri)
    w = wq[None, :] * det[:, None]
    g = np.einsum("qai,eik->eqak", dphi, gl)
    mass = np.einsum("eq,qa,qb->eab", w, phi, phi)
    stiff = np.einsum("eq,eqak,eqbk->eab", w, g, g)
    rows = np.repeat(tri, 6, axis=1).ravel()
    cols = np.tile(tri, (1, 6)).ravel()
    n = mesh.n_nodes
    A = sp.coo_matrix((stiff.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    M = sp.coo_matrix((mass.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    A.sort_indices()
    M.sort_indices()
    X, Y = mesh.node_xy()
    bnd = np.flatnonzero(np.isclose(X, 0) | np.isclose(X, a) | np.isclose(Y, 0) | np.isclose(Y, b))
    return _apply_dirichlet(A, bnd), _apply_dirichlet(M, bnd), bnd


def membrane_analytic(num: int, a: float = 2.0, b: float = 4.0) -> np.ndarray:
    """lambda_mn = pi^2 (m^2/a^2 + n^2/b^2), ``tests/benchmark/vibrating_membrane.py:128-140``."""
    lim = int(np.ceil(np.sqrt(num) * 1.5))
    vals = sorted((np.pi**2) * (m * m / a**2 + n * n / b**2) for m in range(1, lim + 1) for n in range(1, lim + 1))
    return np.array(vals[:num])
