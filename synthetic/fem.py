"""Synthetic (A, M) generator (numpy, CPU): the deterministic inputs of tests, ``bench.py`` and the examples.

Input generation only: no solver arithmetic.  The oracle (``oracle/``, the CPU checker) re-exports it as
``oracle.fem``; the solver package ``lsa-fw_amd/Solver`` never imports it.

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
