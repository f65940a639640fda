// Device-side data model of the mjhip batched stepper.
//
//   DModel : read-only model constants, replicated per GPU, shared by every environment.  Built
//            once from an mjModel by mjh_host.cpp (direct copies of mjModel arrays + tables that
//            are precomputed from it: tree levels, dof-ancestor masks, the static candidate
//            geom-pair list in reference contact order with mixed contact parameters, effective
//            damping/armature).
//   DBatch : per-environment mjData mirror, laid out SoA-across-envs: every mjData field f is ONE
//            device array [nenv][n_f] (environment-major, element innermost) so that the 64
//            lanes of the wavefront that owns environment e read contiguous memory, and so a
//            whole field (e.g. qpos of all envs) is one dense matrix for the caller.
//
// Field lists are X-macros: the host uploader, the device structs and the test accessors are all
// generated from them.
#pragma once

#include "mjh_math.h"

// ---- model: int arrays ------------------------------------------------------------------------
// X(name, count-expression in terms of the size fields of DModel (s.))
#define MJH_MODEL_INT_FIELDS(X)                \
  X(body_parentid, s.nbody)                    \
  X(body_rootid, s.nbody)                      \
  X(body_weldid, s.nbody)                      \
  X(body_simple, s.nbody)                      \
  X(body_mocapid, s.nbody)                     \
  X(body_jntnum, s.nbody)                      \
  X(body_jntadr, s.nbody)                      \
  X(body_dofnum, s.nbody)                      \
  X(body_dofadr, s.nbody)                      \
  X(body_sameframe, s.nbody)                   \
  X(body_level_adr, s.nlevel + 1)              \
  X(body_level_ids, s.nbody)                   \
  X(body_child_adr, s.nbody + 1)               \
  X(body_child_ids, s.nbody)                   \
  X(body_geomnum, s.nbody)                     \
  X(body_geomadr, s.nbody)                     \
  X(body_dofanc, s.nbody * s.nvw)              \
  X(body_treeid, s.nbody)                      \
  X(dof_treeid, s.nv)                          \
  X(tree_dofadr, s.ntree)                      \
  X(tree_dofnum, s.ntree)                      \
  X(jnt_type, s.njnt)                          \
  X(jnt_qposadr, s.njnt)                       \
  X(jnt_dofadr, s.njnt)                        \
  X(jnt_bodyid, s.njnt)                        \
  X(jnt_limited, s.njnt)                       \
  X(jnt_freebody, s.njnt)                      \
  X(M_rowid, s.nC)                             \
  /* the fully implicit integrator (mj_implicit; sizes 0 otherwise): qDeriv's pattern D (rows = dofs: the dofs of the row's  \
     body chain and subtree; D_mapM: the entry's index in M's lower triangle or -1, mju_gatherMasked), the body-by-dof        \
     pattern B of mjd_rne_vel's work arrays (row = ancestors' dofs, own dofs, subtree's dofs), B_pmap: position of an entry \
     in the PARENT's row (-1: the parent is welded to the world, addToParent returns), B_ncopy: leading entries a row        \
     takes over from the parent's (copyFromParent) */ \
  X(D_rowadr, (s.nD ? s.nv : 0))               \
  X(D_rownnz, (s.nD ? s.nv : 0))               \
  X(D_diag, (s.nD ? s.nv : 0))                 \
  X(D_colind, s.nD)                            \
  X(D_rowid, s.nD)                             \
  X(D_mapM, s.nD)                              \
  X(B_rowadr, (s.nD ? s.nbody : 0))            \
  X(B_rownnz, (s.nD ? s.nbody : 0))            \
  X(B_ncopy, (s.nD ? s.nbody : 0))             \
  X(B_pmap, s.nB)                              \
  X(dof_bodyid, s.nv)                          \
  X(dof_jntid, s.nv)                           \
  X(dof_parentid, s.nv)                        \
  X(dof_simplenum, s.nv)                       \
  X(dof_jnttype, s.nv)                         \
  X(M_rownnz, s.nv)                            \
  X(M_rowadr, s.nv)                            \
  X(M_colind, s.nC)                            \
  /* strict lower triangle of M by COLUMN: for dof t, the addresses (into M / M_colind) of the entries M[i][t], i > t, \
     in ascending row order -- the order in which mju_mulSymVecSparse adds the upper-triangle terms to res[t] */ \
  X(M_cscadr, s.nv + 1)                        \
  X(M_cscind, s.nC)                            \
  X(geom_type, s.ngeom)                        \
  X(geom_bodyid, s.ngeom)                      \
  X(geom_sameframe, s.ngeom)                   \
  X(site_bodyid, s.nsite)                      \
  X(site_sameframe, s.nsite)                   \
  X(site_type, s.nsite)                        \
  X(tendon_adr, s.ntendon)                     \
  X(sensor_type, s.nsensor)                    \
  X(sensor_datatype, s.nsensor)                \
  X(sensor_objtype, s.nsensor)                 \
  X(sensor_objid, s.nsensor)                   \
  X(sensor_reftype, s.nsensor)                 \
  X(sensor_refid, s.nsensor)                   \
  X(sensor_dim, s.nsensor)                     \
  X(sensor_adr, s.nsensor)                     \
  X(sensor_intprm0, s.nsensor)                 \
  X(sensor_intprm1, s.nsensor)                 \
  X(sensor_needstage, s.nsensor)               \
  X(geom_rayskip, s.ngeom)                     \
  X(tendon_num, s.ntendon)                     \
  X(tendon_limited, s.ntendon)                 \
  X(ten_J_rownnz, s.ntendon)                   \
  X(ten_J_rowadr, s.ntendon)                   \
  X(ten_J_colind, s.nJten)                     \
  X(wrap_type, s.nwrap)                        \
  X(wrap_objid, s.nwrap)                       \
  X(actuator_trntype, s.nu)                    \
  X(actuator_trnid, 2 * s.nu)                  \
  X(actuator_gaintype, s.nu)                   \
  X(actuator_biastype, s.nu)                   \
  X(actuator_ctrllimited, s.nu)                \
  X(actuator_forcelimited, s.nu)               \
  X(actuator_dyntype, s.nu)                    \
  X(actuator_actadr, s.nu)                     \
  X(actuator_actlimited, s.nu)                 \
  X(actuator_actearly, s.nu)                   \
  X(actuator_disabled, s.nu)  /* mj_actuatorDisabled: the actuator's group is in opt.disableactuator */ \
  X(tendon_actfrclimited, s.ntendon)           \
  X(actuator_momentadr, s.nu + 1)              \
  /* scalar joint transmissions by dof (ascending actuator ids): qfrc_actuator without the search over all rows */ \
  X(dof_act_adr, s.nv + 1)                     \
  X(dof_act_ids, s.nu + 1)                     \
  X(jnt_actfrclimited, s.njnt)                 \
  /* dofs whose joint takes its gravity compensation through qfrc_actuator (jnt_actgravcomp) */ \
  X(dof_actgravcomp, s.nv_actgc)               \
  /* cameras that sensors refer to (frame sensors on cameras, camprojection; sizes 0 otherwise): mj_camlight's inputs */ \
  X(cam_bodyid, s.ncam_s)                      \
  X(cam_mode, s.ncam_s)                        \
  X(cam_targetbodyid, s.ncam_s)                \
  /* 1: some geom of the body uses the ellipsoid fluid model (the inertia-box model is then off for the body) */ \
  X(body_ellipsoid, (s.ngeom_fluid ? s.nbody : 0)) \
  X(pair_geom1, (s.npair + s.nflexpair))                       \
  X(pair_geom2, (s.npair + s.nflexpair))                       \
  X(pair_dim, (s.npair + s.nflexpair))                         \
  X(pair_maxcon, s.npair)                      \
  X(pair_func, s.npair)                        \
  /* broad / midphase emulation (stage_broadphase): index pair into the collidable-body list for  \
     the sweep-and-prune test (lo | hi<<16, -1: not subject to it), route (0 predefined pair, 1   \
     direct, 2 BVH midphase), and for route 2 the static chain of BVH node pairs from the roots   \
     to the pair's leaves (node indices into bvh_aabb) */                                          \
  X(pair_sap, s.npair)                         \
  X(pair_route, s.npair)                       \
  X(pair_mid_adr, s.npair + 1)                 \
  X(pair_mid, 2 * s.nmid)                      \
  X(bp_body, s.nbp)                            \
  /* equality constraints (connect / weld / joint / tendon) */ \
  X(eq_type, s.neq)                            \
  X(eq_obj1id, s.neq)                          \
  X(eq_obj2id, s.neq)                          \
  X(eq_objsite, s.neq)                         \
  X(eq_active0, s.neq)                         \
  X(eq_rowadr, s.neq + 1)                      \
  /* PGS block visitation orders: engine_solver.c shuffles with a PCG32 that is re-seeded at every \
     solver call, so the order array after iteration k depends only on (nefc, k): precomputed */ \
  X(pgs_order_adr, 130)                        \
  /* L'DL fast path: strict-ancestor bit mask of every dof (2 words), and the flattened update \
     list of mj_factorI: for pivot row k, items dst | src<<10 | scl<<20 (indices into qLD) */ \
  X(dof_ancmask, 2 * s.nv)                     \
  X(ld_prog_adr, s.nv + 1)                     \
  X(ld_prog, s.nldprog)                        \
  /* dofs whose row of M has off-diagonal entries, ascending (the generic L'DL routines take only these in sequence) */ \
  X(ld_rows, s.nldrows)                        \
  /* explicit-index rows (mjh_csr.h): the dof chain of every body -- dofs of its weld body, then ancestors, descending -- as a table */ \
  X(body_chainadr, s.nbody + 1)                \
  X(body_chain, s.nchain)                      \
  X(pgs_order, s.npgsorder)               \
  /* convex meshes (mjh_convex.h): hull graph, polygons (mjModel mesh_*, include/mujoco/mjmodel.h:1040-1075) */ \
  X(geom_dataid, s.ngeom)                      \
  X(mesh_vertadr, s.nmesh)                     \
  X(mesh_vertnum, s.nmesh)                     \
  X(mesh_graphadr, s.nmesh)                    \
  X(mesh_polynum, s.nmesh)                     \
  X(mesh_polyadr, s.nmesh)                     \
  X(mesh_graph, s.nmeshgraph)                  \
  X(mesh_extrema, 27 * s.nmesh)                \
  X(mesh_polyvertadr, s.nmeshpoly)             \
  X(mesh_polyvertnum, s.nmeshpoly)             \
  X(mesh_polyvert, s.nmeshpolyvert)            \
  X(mesh_polymapadr, s.nmeshvert)              \
  X(mesh_polymapnum, s.nmeshvert)              \
  X(mesh_polymap, s.nmeshpolymap)              \
  /* flexes (mjh_flex.h): mjModel flex_* tables (include/mujoco/mjmodel.h:541-624) with vertex / edge / element ids \
     made global over all flexes, plus the gather tables that let every dof sum its force contributions in the    \
     reference's order: elements of a vertex (elem<<2 | corner, ascending element), bending edges of a vertex      \
     (edge<<2 | slot), non-zeros of flexedge_J by column (entry index, ascending edge) */                          \
  X(flex_dim, s.nflex)                         \
  X(flex_vertadr, s.nflex)                     \
  X(flex_vertnum, s.nflex)                     \
  X(flex_edgeadr, s.nflex)                     \
  X(flex_edgenum, s.nflex)                     \
  X(flex_elemadr, s.nflex)                     \
  X(flex_elemnum, s.nflex)                     \
  X(flex_rigid, s.nflex)                       \
  X(flex_centered, s.nflex)                    \
  X(flex_stiffnessadr, s.nflex)                \
  X(flex_bendingadr, s.nflex)                  \
  X(flexvert_flex, s.nflexvert)                \
  X(flexvert_bodyid, s.nflexvert)              \
  X(flexedge_flex, s.nflexedge)                \
  X(flexedge_vert, 2 * s.nflexedge)            \
  X(flexedge_flap, 2 * s.nflexbend)            \
  X(flexedge_rigid, s.nflexedge)               \
  /* flex edge equality constraints (mjEQ_FLEX): 1 where a flex's edges are constrained; the edge of every row of   \
     such an equality, indexed like the rows eq_rowadr hands out */                                              \
  X(flex_edgeequality, s.nflex)                \
  X(eqrow_edge, s.neqrow)                      \
  X(flexedge_J_rownnz, s.nflexedge)            \
  X(flexedge_J_rowadr, s.nflexedge)            \
  X(flexedge_J_colind, s.nJfe)                 \
  X(flexedge_J_rowid, s.nJfe)                  \
  X(flexJ_cscadr, s.nflexdof + 1)              \
  X(flexJ_cscind, s.nJfe)                      \
  X(flexJ_cscedge, s.nJfe)                     \
  X(flexelem_flex, s.nflexelem)                \
  X(flexelem_vert, 4 * s.nflexelem)            \
  X(flexelem_edge, 6 * s.nflexelem)            \
  X(flexvert_elemadr, s.nflexvert + 1)         \
  X(flexvert_elem, s.nflexelemdata)            \
  X(flexvert_bendadr, s.nflexvert + 1)         \
  /* vertices riding on articulated bodies (not three axis-aligned sliders of their own): for every dof, the vertices whose \
     body chain holds it, in vertex order -- the order in which mj_flexPassiveStretch's mj_applyFT calls add to that dof */ \
  X(flexdof_vadr, s.nflexdof + 1)              \
  /* interpolated flexes (flex_interp 1 trilinear / 2 quadratic; mj_flex :580-626, mj_flexPassiveInterp): nodes are bodies, \
     vertices are interpolated from the nodes of their cell.  flexnode_*: nodes (global ids),                                \
     flexcell_*: the finite cells (flex, address of the cell's stiffness matrix or -1 for an empty cell, its nodes in \
     mju_flexGatherCellState's order, global ids), flexnode_cell*: for every node the (cell << 5 | local node) items in \
     the order the cells scatter their forces into it */ \
  X(flex_interp, s.nflex)                      \
  X(flex_cellnum, 3 * s.nflex)                 \
  X(flex_nodeadr, s.nflex + 1)                 \
  X(flexnode_bodyid, s.nflexnode)              \
  X(flexnode_flex, s.nflexnode)                \
  X(flexcell_flex, s.nflexcell)                \
  X(flexcell_kadr, s.nflexcell)                \
  X(flexcell_node, 27 * s.nflexcell)           \
  X(flexnode_celladr, s.nflexnode + 1)         \
  X(flexnode_cell, s.nflexnodecell)            \
  /* implicit effective metric M + K (mjh_effmetric.h; mj_flexCG, engine_forward.c:1640; mjd_effBuild, engine_derivative.c:3414): \
     the per-step stiffness K of the standard flexes in dof-level CSR -- structure static (mjd_flexStiff_assemble :1810-2073: \
     vertex slots with three sliders, neighbours through bending flaps and element cliques, sorted by dof address) --; \
     efmblk_*: the 3 x 3 blocks in storage order (slot, neighbour position) with the vertices (global ids) they couple and \
     their contributions in the reference's order of accumulation (kind 0 bending / 1 stretch | edge or element << 1, \
     corner pair i << 2 | j packed in efmblk_cij) */ \
  X(efm_rownnz, s.nefmrow)                     \
  X(efm_rowadr, s.nefmrow)                     \
  X(efm_colind, s.nefmK)                       \
  X(efm_slotvert, s.nefmslot)                  \
  X(efm_slotdiag, s.nefmslot)                  \
  X(efm_vertslot, s.nefmvert)                  \
  X(efmblk_slot, s.nefmblk)                    \
  X(efmblk_pos, s.nefmblk)                     \
  X(efmblk_cadr, s.nefmblk + 1)                \
  X(efmblk_c, s.nefmcon)                       \
  X(efmblk_cij, s.nefmcon)                     \
  /* flex vertex equality constraints (mjEQ_FLEXVERT; mj_flex, engine_core_smooth.c:745-918): the two rows of every \
     vertex of a shell flex with edge equality "vert" (mjModel flexvert_J_*: both rows share one pattern, the dofs of the \
     vertex body and of its edge neighbours), and the vertex -> adjacent edges lists (local edge ids) */ \
  X(fv_rownnz, 2 * s.nfv)                      \
  X(fv_rowadr, 2 * s.nfv)                      \
  X(fv_colind, s.nJfv2)                        \
  X(fv_edgeadr, s.nfv)                         \
  X(fv_edgenum, s.nfv)                         \
  X(fv_edge, s.nfvedge)                        \
  /* bending-only metric (s.efm == 2): the CONSTANT sparse factor of M + (h^2 + h damping) K_bend from mj_setConst \
     (mjModel efm0_*, engine_setconst.c:1366): row -> dof, the factor's rows (off-diagonal entries ascending, diagonal \
     last), its off-diagonal entries by column with the rows descending (e0_cscind: address of the entry, e0_cscrow: its \
     row), and the two level schedules of mju_cholSolveSparse's sweeps (rows whose operands are final, level by level) */ \
  X(e0_dof, s.ne0)                             \
  X(e0_rownnz, s.ne0)                          \
  X(e0_rowadr, s.ne0)                          \
  X(e0_colind, s.ne0L)                         \
  X(e0_cscadr, s.ne0 + 1)                      \
  X(e0_cscind, s.ne0off)                       \
  X(e0_cscrow, s.ne0off)                       \
  X(e0_l1adr, s.ne0lev1 + 1)                   \
  X(e0_l1row, s.ne0)                           \
  X(e0_l2adr, s.ne0lev2 + 1)                   \
  X(e0_l2row, s.ne0)                           \
  X(e0_cov, s.nefmrow)                         \
  X(flexdof_vert, s.nflexdofv)                 \
  X(flexvert_bend, 4 * s.nflexbend)            \
  /* flex collisions (mjh_flexcol.h).  colseg: the collision pass as segments (end of a range of static geom pairs, then \
     the body : flex job that follows it in the reference's bodyflex order, -1 -1 for none); flexjob_*: the geoms of \
     each job's body (their parameter records are pairs npair + k); flexleaf_elem: the elements in every flex's BVH \
     (the active layers), in the order a depth-first walk of the tree visits its leaves */ \
  X(colseg, 3 * s.ncolseg)                     \
  X(flexjob_adr, s.ncolseg + 1)                \
  X(flexjob_geom, s.nflexpair)                 \
  X(flexjob_nsub, s.nflexpair)                 \
  X(flex_leafadr, s.nflex + 1)                 \
  /* flex self-collisions (mjh_flexcol.h: flex_self_collide).  flexself_*: the flexes that collide with themselves (flex, \
     parameter record = pair index, mode 1 sweep-and-prune / 2 all pairs); flexact_*: the active elements of every flex in \
     element order (mj_isElemActive); flexbvh_*: every flex's bounding volume hierarchy -- node range of the flex, children \
     (node ids of this table, -1), element of a leaf (global id, -1 for inner nodes), and the inner nodes ordered by height \
     (children before parents: mj_updateDynamicBVH recomputes the inner boxes bottom-up) */ \
  /* pairs of different flexes that collide: the two flexes, the pair's parameter record */ \
  X(flexff_flex, 2 * s.nflexff)                \
  X(flexff_pair, s.nflexff)                    \
  /* body : flex jobs whose body holds several geoms (its hierarchy has inner nodes): the order in which mj_collideTree's \
     walk meets the (geom, element) pairs matters when more than mjMAXCONPAIR contacts are thinned out.  flexjob_leaf: the \
     node of the job geom's leaf (-1: single-geom body, static order), jobbvh_*: the bodies' nodes (children, parent) */ \
  X(flexjob_leaf, s.nflexpair)                 \
  X(jobbvh_child, 2 * s.njobbvh)               \
  X(jobbvh_parent, s.njobbvh)                  \
  X(flexff_mode, s.nflexff)                    \
  X(flexbvh_parent, s.nflexbvh)                \
  X(flexelem_bvhleaf, ((s.nflexff || s.njobbvh || s.nselfbvh) ? s.nflexelem : 0)) \
  X(flexself_flex, s.nflexself)                \
  X(flexself_pair, s.nflexself)                \
  X(flexself_mode, s.nflexself)                \
  X(flexact_adr, s.nflex + 1)                  \
  X(flexact_elem, s.nflexact)                  \
  X(flexbvh_adr, s.nflex + 1)                  \
  X(flexbvh_child, 2 * s.nflexbvh)             \
  X(flexbvh_elem, s.nflexbvh)                  \
  X(flexbvh_hadr, s.nflexbvhh + 1)             \
  X(flexbvh_order, s.nflexbvh)                 \
  X(flexleaf_elem, s.nflexleaf)                \
  X(flex_mintree, s.nflex)                     \
  X(flex_contype, s.nflex)                     \
  X(flex_conaffinity, s.nflex)                 \
  X(geom_contype, s.ngeomflex)                 \
  X(geom_conaffinity, s.ngeomflex)

// ---- model: real arrays -----------------------------------------------------------------------
#define MJH_MODEL_REAL_FIELDS(X)               \
  X(qpos0, s.nq)                               \
  X(qpos_spring, s.nq)                         \
  X(body_pos, 3 * s.nbody)                     \
  X(body_quat, 4 * s.nbody)                    \
  X(body_ipos, 3 * s.nbody)                    \
  X(body_iquat, 4 * s.nbody)                   \
  X(body_mass, s.nbody)                        \
  X(body_subtreemass, s.nbody)                 \
  X(body_gravcomp, s.nbody)                    \
  X(sensor_cutoff, s.nsensor)                  \
  X(body_inertia, 3 * s.nbody)                 \
  X(body_invweight0, 2 * s.nbody)              \
  X(jnt_pos, 3 * s.njnt)                       \
  X(jnt_axis, 3 * s.njnt)                      \
  X(jnt_stiffness, s.njnt)                     \
  X(jnt_stiffnesspoly, 2 * s.njnt)             \
  X(jnt_range, 2 * s.njnt)                     \
  X(jnt_margin, s.njnt)                        \
  X(jnt_solref, 2 * s.njnt)                    \
  X(jnt_solimp, 5 * s.njnt)                    \
  X(jnt_actfrcrange, 2 * s.njnt)               \
  X(dof_armature_eff, s.nv)                    \
  X(dof_damping_eff, s.nv)                     \
  X(dof_dampingpoly_eff, 2 * s.nv)             \
  X(dof_invweight0, s.nv)                      \
  X(dof_M0, s.nv)                              \
  X(dof_frictionloss, s.nv)                    \
  X(dof_solref, 2 * s.nv)                      \
  X(dof_solimp, 5 * s.nv)                      \
  X(geom_pos, 3 * s.ngeom)                     \
  X(geom_quat, 4 * s.ngeom)                    \
  X(geom_size, 3 * s.ngeom)                    \
  /* ellipsoid fluid model (geoms with fluidshape = ellipsoid; sizes 0 otherwise): the twelve interaction coefficients of a  \
     geom (mjNFLUID) and its semi-axes (mju_geomSemiAxes) */ \
  X(cam_pos, 3 * s.ncam_s)                     \
  X(cam_quat, 4 * s.ncam_s)                    \
  X(cam_mat0, 9 * s.ncam_s)                    \
  X(cam_pos0, 3 * s.ncam_s)                    \
  X(cam_poscom0, 3 * s.ncam_s)                 \
  /* camprojection: focal lengths in pixels (fx, fy) and the resolution (cam_project, engine_sensor.c:281-316; model constants) */ \
  X(cam_proj, 4 * s.ncam_s)                    \
  X(geom_fluid, 12 * s.ngeom_fluid)            \
  X(geom_semiaxes, 3 * s.ngeom_fluid)          \
  X(geom_rbound, s.ngeom)                      \
  X(geom_aabb, 6 * s.ngeom)                    \
  X(geom_bpmargin, s.ngeom)                    \
  X(bvh_aabb, 6 * s.nbvh)                      \
  X(pair_bodymargin, s.npair)                  \
  X(body_bpext, s.nbody)                       \
  X(site_pos, 3 * s.nsite)                     \
  X(site_quat, 4 * s.nsite)                    \
  X(site_size, 3 * s.nsite)                    \
  X(geom_surfacevel, 6 * s.ngeom)              \
  X(tendon_range, 2 * s.ntendon)               \
  X(tendon_actfrcrange, 2 * s.ntendon)         \
  X(tendon_margin, s.ntendon)                  \
  X(tendon_solref_lim, 2 * s.ntendon)          \
  X(tendon_solimp_lim, 5 * s.ntendon)          \
  X(tendon_solref_fri, 2 * s.ntendon)          \
  X(tendon_solimp_fri, 5 * s.ntendon)          \
  X(tendon_invweight0, s.ntendon)              \
  X(tendon_stiffness, s.ntendon)               \
  X(tendon_stiffnesspoly, 2 * s.ntendon)       \
  X(tendon_damping_eff, s.ntendon)             \
  X(tendon_dampingpoly_eff, 2 * s.ntendon)     \
  X(tendon_armature_eff, s.ntendon)            \
  X(tendon_lengthspring, 2 * s.ntendon)        \
  X(tendon_frictionloss, s.ntendon)            \
  X(wrap_prm, s.nwrap)                         \
  X(actuator_gear, 6 * s.nu)                   \
  X(actuator_ctrlrange, 2 * s.nu)              \
  X(actuator_forcerange, 2 * s.nu)             \
  X(actuator_actrange, 2 * s.nu)               \
  X(actuator_dyntau, s.nu)                     \
  X(actuator_dynprm, 3 * s.nu)   /* muscle dynamics: tau_act, tau_deact, smoothing width */ \
  X(actuator_lengthrange, 2 * s.nu)            \
  X(actuator_acc0, s.nu)                       \
  X(actuator_gainprm, 10 * s.nu)               \
  X(actuator_biasprm, 10 * s.nu)               \
  X(actuator_cranklength, s.nu)                \
  X(pair_margin, (s.npair + s.nflexpair))                      \
  X(pair_includemargin, (s.npair + s.nflexpair))               \
  X(pair_adhesion, (s.npair + s.nflexpair))   /* mj_contactParam's adhesion of the pair (0: none) */ \
  X(pair_friction, 5 * (s.npair + s.nflexpair))                \
  X(pair_solref, 2 * (s.npair + s.nflexpair))                  \
  X(pair_solreffriction, 2 * (s.npair + s.nflexpair))          \
  X(pair_solimp, 5 * (s.npair + s.nflexpair))                  \
  X(eq_solref, 2 * s.neq)                      \
  X(eq_solimp, 5 * s.neq)                      \
  X(eq_data, 11 * s.neq)                       \
  X(tendon_length0, s.ntendon)                 \
  /* mesh vertices (float in mjModel; widened exactly) and polygon normals */ \
  X(mesh_vert, 3 * s.nmeshvert)                \
  X(mesh_polynormal, 3 * s.nmeshpoly)          \
  /* flexes */                                 \
  X(flex_vert, 3 * s.nflexvert)                \
  /* interpolated flexes: node offsets in their bodies' frames, node positions in qpos0, the vertices' parametric coordinates */ \
  X(jobbvh_surface, s.njobbvh)                 \
  X(flex_node, 3 * s.nflexnode)                \
  X(flex_node0, 3 * s.nflexnode)               \
  X(flex_vert0, 3 * s.nflexivert)              \
  X(flexedge_length0, s.nflexedge)             \
  X(flexedge_invweight0, s.nflexedge)          \
  X(flex_stiffness, s.nflexstiffness)          \
  X(flex_bending, s.nflexbending)              \
  X(e0_L, s.ne0L)                              \
  /* flex vertex constraints: inverse reference shape matrix of every vertex (flex_vertmetric), rest vector of every \
     edge from its first to its second vertex scaled by the flex's bounding box (2 flex_size) */ \
  X(fv_metric, 4 * s.nfv)                      \
  X(fv_dx, 3 * s.nfvdx)                        \
  X(flex_damping, s.nflex)                     \
  X(flex_edgestiffness, s.nflex)               \
  X(flex_edgedamping, s.nflex)                 \
  X(flex_radius, s.nflex)                      \
  /* edge spring / damper coefficients of every edge's flex (0 for rigid edges and rigid flexes) */ \
  X(flexedge_k, s.nflexedge)                   \
  X(flexedge_d, s.nflexedge)

// ---- compile-time feature set of a kernel variant -------------------------------------------------
// The stage sources are compiled several times (mjh_modes.h); each compilation defines MJH_FEATURES,
// the set of optional model features its kernels carry.  A heavy branch is written
// `if (MJH_HAS(MJH_FT_x) && <run-time condition>)`, so a lean variant does not pay -- in code size,
// stack frame, registers -- for features its models never use.  DSizes::features is the set a model
// NEEDS (mjh_model_build.h); the runtime only launches variants whose set covers it.
enum {
  MJH_FT_PRIMAL        = 1<<0,   // Newton / CG solvers
  MJH_FT_ELLIPTIC      = 1<<1,   // elliptic friction cones
  MJH_FT_EQUALITY      = 1<<2,   // equality constraints
  MJH_FT_RK4           = 1<<3,
  MJH_FT_IMPLICIT      = 1<<4,   // implicitfast integrator
  MJH_FT_SENSOR        = 1<<5,
  MJH_FT_COLCONVEX     = 1<<6,   // colliders beyond plane / sphere / capsule (box, cylinder, ...)
  MJH_FT_CONDIM46      = 1<<7,   // torsional / rolling friction rows
  MJH_FT_ACT           = 1<<8,   // stateful actuators (na > 0)
  MJH_FT_TENDONSPATIAL = 1<<9,   // spatial tendons
  MJH_FT_TRNMISC       = 1<<10,  // transmissions other than joint / jointinparent on slide or hinge joints
  MJH_FT_PASSIVEMISC   = 1<<11,  // gravity compensation, fluid forces, surface velocities, polynomial springs / dampers
  MJH_FT_MOCAP         = 1<<12,
  MJH_FT_ISLANDS       = 1<<13,  // more than one kinematic tree (union-find; per-island solves)
  MJH_FT_GAINBIAS      = 1<<14,  // affine gains / biases, force / act limits, joint actuator-force limits
  MJH_FT_FLEX          = 1<<15,  // flex objects (mjh_flex.h)
  MJH_FT_ALL           = 0x7fffffff,
  MJH_FT_LEAN          = 0,
};
#define MJH_HAS(f) (((MJH_FEATURES) & (f)) != 0)

struct DSizes {
  int nq, nv, nu, na, nbody, njnt, ngeom, nsite, ntendon, nwrap, nC, nJten, ntree;
  int ncam_s;          // ncam when some sensor is attached to / projects into a camera, else 0
  int ngeom_fluid;     // ngeom when some geom uses the ellipsoid fluid model, else 0
  int nv_actgc;        // nv when some joint has actuator-level gravity compensation, else 0
  int nD, nB;          // the fully implicit integrator: entries of qDeriv's pattern, of the body-by-dof pattern (else 0)
  int features;    // MJH_FT_* bits this model needs from a kernel variant
  int neq;         // equality constraints
  int nlevel;      // depth levels of the kinematic tree (world = level 0)
  int nvw;         // 32-bit words per dof-ancestor mask = (nv+31)/32
  int npair;       // static candidate geom pairs (reference contact order)
  int nbp;         // collidable bodies (sweep-and-prune participants), 0: no pair is subject to the broadphase cull
  int nbvh, nmid;  // static BVH nodes; entries of the midphase chains
  int npassw;      // 32-bit words of the per-step pair pass mask
  int bp_any_mid;  // some pair takes the BVH midphase route
  int bp_any_sap;  // some pair is subject to the sweep-and-prune cull (else only midphase tests run)
  int nmoment;     // capacity of the sparse actuator_moment (sum of per-actuator row capacity)
  int nconmax;     // per-env contact capacity
  int nconlds;     // contact slots kept in LDS by the residency plan
  int nconH;       // contacts with a cone-Hessian slot (elliptic cones + primal solver), else 0
  int nefcmax;     // per-env constraint-row capacity
  int nefcAR;      // rows of the dense efc_AR home: nefcmax under the dual solver (PGS), 0 under the primal ones (never built)
  int nstate;      // mj_stateSize(FULLPHYSICS)
  int nsensor, nsensordata;
  int nmocap;
  int nuserdata;
  int nbody_fluid;  // nbody when fluid forces are on, else 0
  int nbody_sens;  // nbody when the model has sensors (cacc / cfrc / subtree velocity arrays), else 0
  int sens_rnepost, sens_subtreevel;   // some sensor needs mj_rnePostConstraint / mj_subtreeVel
  int sens_energy;     // bit 0: a potential-energy sensor (mj_energyPos), bit 1: a kinetic-energy sensor (mj_energyVel, reads M's global home)
  int npgsorder;   // entries of the precomputed PGS visitation-order table
  int nldprog;     // entries of the flattened L'DL update list
  int nldrows;     // dofs whose row of M has off-diagonal entries
  int nchain;      // entries of the body dof-chain table (explicit-index rows; 1 otherwise)
  int ld_fast;     // 1: the register-resident L'DL routines apply (nv <= 64, nC <= 1024, depth <= 16)
  int pgs_iters;   // iterations covered by that table (min(opt.iterations, 128))
  int pgs_nmax;    // largest nefc covered by that table (64, or 128 when the constraint capacity allows more than 64 rows)
  // convex collision (mjh_convex.h): mesh table sizes and the per-lane GJK / EPA workspace
  int nmesh, nmeshvert, nmeshgraph, nmeshpoly, nmeshpolyvert, nmeshpolymap;
  int ccd_any;         // some static pair uses the GJK / EPA narrowphase
  int ccd_N;           // opt.ccd_iterations
  int ccd_P, ccd_D;    // max(npolygonmax, 4), max(nmeshdegmax, 3)
  // row workspace of the convex narrowphase (mjh_convex.h; all 0 without convex pairs): reals of a row's fast page,
  // reals + ints of it counted in reals (the LDS-planned field ccd_row holds 4 of these per environment), bytes of a
  // row's overflow page and of an environment's global block (header, contact records, 4 overflow pages)
  int ccd_row_freal, ccd_row_reals, ccd_slow_bytes, ccd_env_bytes;
  int ccd_npoly;       // static pairs that may return several contacts (box / mesh against box / mesh without margin)
  int ccd_rows;        // row workspaces per environment: 4 (one wavefront), 4 MJH_MW for flex models (multi-wavefront workgroups)
  // sparse constraint path (mjh_sparse.h): 1 when the reference runs its sparse code (mj_isSparse; nv <= 128 here);
  // capacity of the CSR Jacobian; entries of the compressed factor (Newton; x2 with cones)
  int sparse, nJmax, nLp, nLpc;
  int nARw;        // 64-bit words per row of the structural pattern of efc_AR (sparse path under the dual solver), else 0
  // flexes (mjh_flex.h): mjModel sizes; nflexbend = nflexedge when some flex has bending stiffness, else 0;
  // nflexdof = nv with flexes, else 0
  int nflex, nflexvert, nflexedge, nflexelem, nflexelemdata, nflexstiffness, nflexbending, nJfe, nflexbend, nflexdof;
  // implicit effective metric (0 unless mj_flexCG holds): rows (nv), stored entries of K, vertex slots, flex vertices (nflexvert),
  // 3 x 3 blocks, contributions
  int efm, nefmrow, nefmK, nefmslot, nefmvert, nefmblk, nefmcon;
  // efm == 2 (bending stiffness only): rows / entries / off-diagonal entries of the constant factor, levels of its two sweeps
  int ne0, ne0L, ne0off, ne0lev1, ne0lev2;
  // flex vertex equality constraints: nflexvert when some flex carries them (else 0), entries of both rows of all vertices
  // (2 nJfv), entries of flex_vertedge, edges with a rest vector (nflexedge)
  int nfv, nJfv2, nfvedge, nfvdx;
  // interpolated flexes: nodes, vertices with interpolation tables (nflexvert when some flex is interpolated, else 0), finite
  // cells, entries of flexnode_cell
  int nflexnode, nflexivert, nflexcell, nflexnodecell;
  int nconside;        // interpolated flexes: capacity of a contact's list of node bodies (both sides), else 0
  int nflexdofv;       // entries of flexdof_vert (0: every vertex body is three sliders of its own, or pinned to the world)
  int flex_sliders;    // 1: every flex vertex body has body_simple 2 or no dofs up to the world (fast paths of mjh_flex.h)
  // flex collisions: geom : flex parameter records, collision segments, BVH leaves, candidate capacity of one
  // body : flex job, ngeom with flexes (else 0), per-env capacity of the flex contact identity table (nconmax or 0)
  int nflexpair, ncolseg, nflexleaf, nflexcand, ngeomflex, nconflex;
  // flex self-collisions: flexes that collide with themselves, active elements, nodes / heights of the flex bounding volume
  // hierarchies (0 unless some flex collides with itself by sweep-and-prune: the sweep axis is read off the root box)
  int nflexself, nflexact, nflexbvh, nflexbvhh;
  int nflexff;         // pairs of different flexes that collide
  int njobbvh;         // nodes of the body hierarchies of multi-geom body : flex jobs
  int nselfbvh;        // flexes whose self-collisions go through their bounding volume hierarchy
#define MJH_CONFLEX 6          // ints of a contact's flex identity: flex, element, vertex of side 1, then of side 0 (-1: a geom)
  // compressed constraint Jacobian with explicit column indices (mjh_csr.h): 1 for models beyond 128 dofs under CG;
  // capacity of one row
  int csr, csr_rowmax;
  // Newton on the explicit-index rows (mjh_newtonx.h): 1 when that path is taken, entries of the packed lower triangle
  // nv (nv + 1) / 2 (the factor; x2 with elliptic cones), 32-bit words of a dof set
  int xn, xncap, xnw, xnell;
#define MJH_CSR_CHAIN_MAX 64   // per-lane array a contact row's merged dof chain is assembled in (model build keeps csr_rowmax below it)
  // rows of the flex edge equality constraints (eq_rowadr[neq] when the model has one, else 0)
  int neqrow;
  // dofs with friction loss / limited joints (0: stage_make_constraint leaves their candidate ranges out of its scans)
  int ndoffric, njntlim;
};

struct DOptions {
  real timestep, impratio, tolerance, ls_tolerance;
  real gravity[3];
  real magnetic[3];
  real meaninertia;
  int integrator, cone, solver, iterations, ls_iterations;
  int noslip_iterations;   // > 0: mj_solNoSlip after the main solver (dense constraint path; efc_AR is built under every solver)
  real noslip_tolerance;
  int disableflags, enableflags;
  int euler_damp;   // 1: mj_EulerSkip takes the implicit-damping branch (engine_forward.c:1409-1420)
  int has_refsite;    // a site transmission with a reference site (reads xquat at the transmission stage)
  int has_ten_armature;
  int has_act_disabled;  // some actuator is disabled through its group (opt.disableactuator)
  int has_ten_actfrc;    // some tendon limits the total force of the actuators acting on it
  int has_tendon_wrap;   // some spatial tendon wraps around a sphere / cylinder: the tendon stage reads the geom frames
  int has_gravcomp;
  int has_surfacevel; // some geom has a surface velocity (conveyor belts)
  int has_adhesion;   // some geom or pair is adhesive (mjModel.flg_adhesion)
  int has_fluid;      // opt.density / opt.viscosity set: inertia-box fluid forces
  real density, viscosity, wind[3];
  real ccd_tolerance;     // opt.ccd_tolerance
  real ccd_sin, ccd_cos;  // sin / cos of 5e-4 (half the multiccd perturbation angle), evaluated by the host's libm
};

// (device build: the table pointers are constant-address-space pointers, so a wave-uniform index
// turns into a scalar load and a per-lane index into a global load with a scalar base)
struct DModel {
  DSizes s;
  DOptions o;
#define X(name, cnt) const MJH_CONST_AS int* name;
  MJH_MODEL_INT_FIELDS(X)
#undef X
#define X(name, cnt) const MJH_CONST_AS real* name;
  MJH_MODEL_REAL_FIELDS(X)
#undef X
};

// ---- batch: per-environment arrays -------------------------------------------------------------
// Residency.  Every field has a global-memory home [nenv][gcnt] (what the C ABI exposes).  A batch
// can additionally carry an LDS plan: each one-wavefront workgroup owns `lds_bytes` of LDS, and a
// field with an LDS offset is read and written THERE while it is live; its global home is touched
// only at kernel boundaries (persistent state) or by the debug write-back.  The plan is computed
// on the host (mjh_runtime.h: plan_lds) from the lifetimes below by interval overlay: two fields
// whose lifetimes do not intersect may share LDS bytes.
//
// step timeline (the order stages run in; lifetimes are [first write, last read] on this axis)
enum {
  // (collision runs right after kinematics: it only needs the geom and inertial frames, which can
  // then leave LDS early -- the contact slots that replace them are smaller -- and its broad / midphase
  // checks find the inertial frames still resident)
  // (the mass matrix is built and factorised only when its first consumer -- the smooth acceleration --
  // is next: qLD does not sit in LDS through the velocity stages, whose per-body spatial vectors
  // (cvel, cacc, cfrc) then fit)
  MJH_T_BEGIN = 0, MJH_T_KIN = 1, MJH_T_COLLISION = 2, MJH_T_COMPOS = 3, MJH_T_TENDON = 4,
  MJH_T_TRANSMISSION = 5, MJH_T_TAVEL = 6, MJH_T_COMVEL = 7, MJH_T_PASSIVE = 8, MJH_T_RNE = 9,
  MJH_T_CRB = 10, MJH_T_FACTOR = 11, MJH_T_ACTUATION = 12, MJH_T_ACCEL = 13,
  MJH_T_MAKE = 14, MJH_T_PROJECT = 15, MJH_T_REFERENCE = 16, MJH_T_CONSTRAINT = 17,
  MJH_T_FINISH = 18, MJH_T_EULER = 19, MJH_T_END = 20,
};
#define MJH_T_GLB (-1)         // global only
#define MJH_LDS_CON (s.nconlds)  // contact slots kept in LDS (the rest of a contact list is global)

// X(name, global per-env count, LDS count, first, last)
//   first == MJH_T_BEGIN && last == MJH_T_END : persistent state, loaded from / stored to its
//   global home at kernel entry / exit; last == MJH_T_END alone: stored at kernel exit (exported)
#define MJH_BATCH_REAL_FIELDS(X)                                                  \
  X(time, 1, 1, MJH_T_BEGIN, MJH_T_END)                                           \
  X(qpos, s.nq, s.nq, MJH_T_BEGIN, MJH_T_END)                                     \
  X(qvel, s.nv, s.nv, MJH_T_BEGIN, MJH_T_END)                                     \
  X(act, s.na, s.na, MJH_T_BEGIN, MJH_T_END)                                      \
  X(act_dot, s.na, s.na, MJH_T_ACTUATION, MJH_T_END)                              \
  X(mocap_pos, 3 * s.nmocap, 0, MJH_T_GLB, MJH_T_GLB)                             \
  /* per body: the fluid wrench (6), then -- for the implicit integrators' velocity derivative, mjd_inertiaBoxFluid -- the       \
     scalar coefficients B of its twelve J'BJ terms: viscous torque, viscous force, quadratic drag on the six local axes (8) */ \
  X(fluid_frc, 14 * s.nbody_fluid, 0, MJH_T_GLB, MJH_T_GLB)                       \
  X(mocap_quat, 4 * s.nmocap, 0, MJH_T_GLB, MJH_T_GLB)                            \
  X(ctrl, s.nu, s.nu, MJH_T_BEGIN, MJH_T_END)                                     \
  X(qfrc_applied, s.nv, s.nv, MJH_T_BEGIN, MJH_T_END)                             \
  X(qacc_warmstart, s.nv, s.nv, MJH_T_BEGIN, MJH_T_END)                           \
  X(xfrc_applied, 6 * s.nbody, 0, MJH_T_GLB, MJH_T_GLB)                           \
  X(userdata, s.nuserdata, 0, MJH_T_GLB, MJH_T_GLB)                               \
  X(xpos, 3 * s.nbody, 3 * s.nbody, MJH_T_KIN, MJH_T_KIN)                         \
  X(xquat, 4 * s.nbody, 4 * s.nbody, MJH_T_KIN, MJH_T_KIN)                        \
  /* (flex forces are rotated into the vertex bodies' frames in the passive stage) */ \
  X(xmat, 9 * s.nbody, 9 * s.nbody, MJH_T_KIN, (s.nflex ? MJH_T_PASSIVE : MJH_T_COMPOS))  \
  X(xipos, 3 * s.nbody, 3 * s.nbody, MJH_T_KIN, MJH_T_COMPOS)                     \
  X(ximat, 9 * s.nbody, 9 * s.nbody, MJH_T_KIN, MJH_T_COMPOS)                     \
  X(xanchor, 3 * s.njnt, 3 * s.njnt, MJH_T_KIN, MJH_T_COMPOS)                     \
  X(xaxis, 3 * s.njnt, 3 * s.njnt, MJH_T_KIN, MJH_T_COMPOS)                       \
  X(geom_xpos, 3 * s.ngeom, 3 * s.ngeom, MJH_T_KIN, MJH_T_COLLISION)              \
  X(geom_xmat, 9 * s.ngeom, 9 * s.ngeom, MJH_T_KIN, MJH_T_COLLISION)              \
  /* row workspaces of the convex narrowphase (mjh_convex.h): ccd_rows x (shape frames, simplex, polytope / clipping buffers); \
     LDS only -- no global home here: when the plan leaves it out, the rows work in the environment's block of ccd_ws */ \
  X(ccd_row, 0, s.ccd_rows * s.ccd_row_reals, MJH_T_COLLISION, MJH_T_COLLISION)  \
  X(site_xpos, 3 * s.nsite, 3 * s.nsite, MJH_T_KIN, MJH_T_TRANSMISSION)             \
  X(site_xmat, 9 * s.nsite, 9 * s.nsite, MJH_T_KIN, MJH_T_TRANSMISSION)             \
  X(subtree_com, 3 * s.nbody, 3 * s.nbody, MJH_T_COMPOS, MJH_T_MAKE)              \
  X(cinert, 10 * s.nbody, 10 * s.nbody, MJH_T_COMPOS, MJH_T_CRB)                  \
  X(cdof, 6 * s.nv, 6 * s.nv, MJH_T_COMPOS, MJH_T_MAKE)                           \
  X(ten_length, s.ntendon, s.ntendon, MJH_T_TENDON, MJH_T_MAKE)                   \
  X(ten_J, s.nJten, s.nJten, MJH_T_TENDON, MJH_T_MAKE)                            \
  X(ten_velocity, s.ntendon, s.ntendon, MJH_T_TAVEL, MJH_T_PASSIVE)               \
  X(actuator_length, s.nu, s.nu, MJH_T_TRANSMISSION, MJH_T_ACTUATION)             \
  X(actuator_moment, s.nmoment, s.nmoment, MJH_T_TRANSMISSION, MJH_T_ACTUATION)   \
  X(actuator_velocity, s.nu, s.nu, MJH_T_TAVEL, MJH_T_ACTUATION)                  \
  X(actuator_force, s.nu, s.nu, MJH_T_ACTUATION, MJH_T_ACTUATION)                 \
  X(crb, 10 * s.nbody, 10 * s.nbody, MJH_T_CRB, MJH_T_CRB)                        \
  /* the mass matrix is assembled in qLD's slot (stage_crb) and copied to this global home: the copy tests \
     inspect and the one mj_Euler / implicitfast / Newton read after qLD has been factorised in place */ \
  X(M, s.nC, 0, MJH_T_GLB, MJH_T_GLB)                                             \
  X(qLD, s.nC, s.nC, MJH_T_CRB, MJH_T_EULER)                                   \
  X(qLDiagInv, s.nv, s.nv, MJH_T_FACTOR, MJH_T_EULER)                             \
  /* factor of qH = M + h*diag(B) when mj_Euler's solve is paired with the finish solve (stage_finish) */ \
  X(qH2, s.nC, s.nC, MJH_T_FINISH, MJH_T_EULER)                                   \
  /* qH while it is factorised next to M (stage_factor_m); parked in qH2's global home until finish */ \
  X(qHtmp, s.nC, s.nC, MJH_T_FACTOR, MJH_T_FACTOR)                                \
  X(qHtmpDiagInv, s.nv, s.nv, MJH_T_FACTOR, MJH_T_FACTOR)                         \
  X(qH2DiagInv, s.nv, s.nv, MJH_T_FINISH, MJH_T_EULER)                            \
  X(cvel, 6 * s.nbody, 6 * s.nbody, MJH_T_COMVEL, MJH_T_RNE)                      \
  X(cdof_dot, 6 * s.nv, 6 * s.nv, MJH_T_COMVEL, MJH_T_RNE)                        \
  X(cacc, 6 * s.nbody, 6 * s.nbody, MJH_T_RNE, MJH_T_RNE)                         \
  X(cfrc, 6 * s.nbody, 6 * s.nbody, MJH_T_RNE, MJH_T_RNE)                         \
  X(qfrc_spring, s.nv, s.nv, MJH_T_PASSIVE, MJH_T_PASSIVE)                        \
  X(qfrc_damper, s.nv, s.nv, MJH_T_PASSIVE, MJH_T_PASSIVE)                        \
  X(qfrc_passive, s.nv, s.nv, MJH_T_PASSIVE, MJH_T_ACCEL)                         \
  X(qfrc_bias, s.nv, s.nv, MJH_T_RNE, MJH_T_ACCEL)                                \
  X(qfrc_actuator, s.nv, s.nv, MJH_T_ACTUATION, MJH_T_ACCEL)                      \
  X(qfrc_smooth, s.nv, s.nv, MJH_T_ACCEL, MJH_T_EULER)                            \
  X(qacc_smooth, s.nv, s.nv, MJH_T_ACCEL, MJH_T_FINISH)                           \
  X(qfrc_constraint, s.nv, s.nv, MJH_T_CONSTRAINT, MJH_T_EULER)                   \
  X(qacc, s.nv, s.nv, MJH_T_CONSTRAINT, MJH_T_END)                                \
  X(qe, s.nv, s.nv, MJH_T_FINISH, MJH_T_EULER)                                     \
  X(con_dist, s.nconmax, MJH_LDS_CON, MJH_T_COLLISION, MJH_T_MAKE)                \
  X(con_pos, 3 * s.nconmax, 3 * MJH_LDS_CON, MJH_T_COLLISION, MJH_T_MAKE)         \
  X(con_frame, 9 * s.nconmax, 9 * MJH_LDS_CON, MJH_T_COLLISION, MJH_T_MAKE)       \
  X(con_mu, s.nconmax, MJH_LDS_CON, MJH_T_COLLISION, MJH_T_MAKE)                  \
  /* flexes: mjData flexvert_xpos / flexedge_length / flexedge_velocity / flexedge_J; per-element and per-bending-edge \
     force blocks before the ordered per-vertex sums (mjh_flex.h) */                 \
  X(flexvert_xpos, 3 * s.nflexvert, 0, MJH_T_GLB, MJH_T_GLB)                      \
  X(flexedge_length, s.nflexedge, 0, MJH_T_GLB, MJH_T_GLB)                        \
  X(flexedge_velocity, s.nflexedge, 0, MJH_T_GLB, MJH_T_GLB)                      \
  X(flexedge_J, s.nJfe, 0, MJH_T_GLB, MJH_T_GLB)                                  \
  X(flexelem_frc, 12 * s.nflexelem, 0, MJH_T_GLB, MJH_T_GLB)                      \
  /* interpolated flexes: node positions / velocities, the cells' node forces (spring | damper, 81 reals each), and per \
     contact the node bodies and weights of an interpolated side (mj_vertBodyWeight) */ \
  X(flexnode_xpos, 3 * s.nflexnode, 0, MJH_T_GLB, MJH_T_GLB)                      \
  X(flexnode_vel, 3 * s.nflexnode, 0, MJH_T_GLB, MJH_T_GLB)                       \
  X(flexcell_in, 166 * s.nflexcell, 0, MJH_T_GLB, MJH_T_GLB)                      \
  X(flexcell_frc, 162 * s.nflexcell, 0, MJH_T_GLB, MJH_T_GLB)                     \
  X(con_nodew, s.nconside * s.nconmax, 0, MJH_T_GLB, MJH_T_GLB)       \
  /* implicit effective metric: K's values, factored diagonal blocks, shift c, per-element stretch blocks / edge terms, PCG vectors */ \
  X(efm_K_val, s.nefmK, 0, MJH_T_GLB, MJH_T_GLB)                                  \
  X(efm_L, 9 * s.nefmslot, 0, MJH_T_GLB, MJH_T_GLB)                               \
  X(efm_c, s.nefmrow, 0, MJH_T_GLB, MJH_T_GLB)                                    \
  X(efm_eblk, (s.efm ? 144 * s.nflexelem : 0), 0, MJH_T_GLB, MJH_T_GLB)           \
  X(efm_work, 6 * s.nefmrow, 0, MJH_T_GLB, MJH_T_GLB)                             \
  X(efm_bz, s.ne0, 0, MJH_T_GLB, MJH_T_GLB)                                       \
  /* mjData flexvert_length, flexvert_J */ \
  X(flexvert_length, 2 * s.nfv, 0, MJH_T_GLB, MJH_T_GLB)                          \
  X(flexvert_J, s.nJfv2, 0, MJH_T_GLB, MJH_T_GLB)                                 \
  X(flexvert_frc, 3 * (s.nflexdofv ? s.nflexvert : 0), 0, MJH_T_GLB, MJH_T_GLB)   \
  X(flexbend_frc, 24 * s.nflexbend, 0, MJH_T_GLB, MJH_T_GLB)                      \
  /* mjData flexelem_aabb; candidate contacts of one body : flex job (dist, pos[3], normal[3], min_dist) */ \
  X(flexelem_aabb, 6 * s.nflexelem, 0, MJH_T_GLB, MJH_T_GLB)                      \
  /* product vectors of the solver's ordered sums when the LDS block has no room for them (mjh_newton.h: csr_dots) */ \
  X(csr_prod, 6 * s.csr * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                          \
  /* Newton on the explicit-index rows: the factor(s) as packed lower triangles, dense row / vector work space */ \
  X(xn_L, s.xn * s.xncap, 0, MJH_T_GLB, MJH_T_GLB)                                \
  X(xn_Lc, s.xnell * s.xncap, 0, MJH_T_GLB, MJH_T_GLB)                             \
  X(xn_rw, 12 * s.xn * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                              \
  X(flexcand, 8 * s.nflexcand, 0, MJH_T_GLB, MJH_T_GLB)                           \
  X(flexbvh_aabb, 6 * s.nflexbvh, 0, MJH_T_GLB, MJH_T_GLB)                        \
  X(efc_J, s.nefcmax * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                             \
  X(efc_Y, s.nefcmax * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                             \
  X(efc_AR, s.nefcAR * s.nefcAR, 0, MJH_T_GLB, MJH_T_GLB)                          \
  X(efc_pos, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                  \
  X(efc_margin, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                               \
  X(efc_frictionloss, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                         \
  X(efc_diagA, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                \
  X(efc_KBIP, 4 * s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                             \
  X(efc_D, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                    \
  X(efc_R, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                    \
  X(efc_vel, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                  \
  X(efc_aref, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                 \
  X(efc_b, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                    \
  X(efc_force, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                \
  X(efc_cone, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                 \
  X(sensordata, s.nsensordata, 0, MJH_T_GLB, MJH_T_GLB)                           \
  X(cacc_post, 6 * s.nbody_sens, 0, MJH_T_GLB, MJH_T_GLB)                         \
  X(cfrc_int, 6 * s.nbody_sens, 0, MJH_T_GLB, MJH_T_GLB)                          \
  X(cfrc_ext, 6 * s.nbody_sens, 0, MJH_T_GLB, MJH_T_GLB)                          \
  X(sens_bvel, 6 * s.nbody_sens, 0, MJH_T_GLB, MJH_T_GLB)                         \
  X(subtree_linvel, 3 * s.nbody_sens, 0, MJH_T_GLB, MJH_T_GLB)                    \
  X(subtree_angmom, 3 * s.nbody_sens, 0, MJH_T_GLB, MJH_T_GLB)                    \
  X(con_H, 36 * s.nconH, 0, MJH_T_GLB, MJH_T_GLB)                                 \
  /* primal Newton solver: dense M, Hessian / Cholesky factor, nv-vectors */       \
  X(nt_M, (1 - s.sparse - s.csr) * s.nv * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                  \
  X(nt_H, (1 - s.sparse - s.csr) * s.nv * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                  \
  X(nt_vec, 8 * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                                    \
  /* sparse constraint path: CSR Jacobian values, its transpose, packed factor L (row r at r(r+1)/2), Lcone */ \
  X(sp_J, s.nJmax, 0, MJH_T_GLB, MJH_T_GLB)                                       \
  X(sp_JT, s.nJmax, 0, MJH_T_GLB, MJH_T_GLB)                                      \
  X(sp_L, s.nLp, 0, MJH_T_GLB, MJH_T_GLB)                                         \
  X(sp_Lc, s.nLpc, 0, MJH_T_GLB, MJH_T_GLB)                                       \
  /* mj_RungeKutta intermediates: X[4] = (qpos, qvel), F[4] = qacc, dX */           \
  X(rk_X, 4 * (s.nq + s.nv), 0, MJH_T_GLB, MJH_T_GLB)                             \
  X(rk_F, 4 * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                                      \
  /* the fully implicit integrator: qDeriv, qLU = M - h qDeriv (factorised in place), and mjd_rne_vel's work arrays */ \
  /* ellipsoid fluid model, per geom: the wrench in the world frame (6), then the 6 x 6 derivative of the local wrench with  \
     respect to the local velocity (implicit integrators; mjd_ellipsoidFluid) */ \
  X(fluid_geom, 42 * s.ngeom_fluid, 0, MJH_T_GLB, MJH_T_GLB)                      \
  /* camera frames (mj_camlight), evaluated with the position-stage sensors */ \
  X(cam_xpos, 3 * s.ncam_s, 0, MJH_T_GLB, MJH_T_GLB)                              \
  X(cam_xmat, 9 * s.ncam_s, 0, MJH_T_GLB, MJH_T_GLB)                              \
  X(qfrc_gravcomp, s.nv_actgc, 0, MJH_T_GLB, MJH_T_GLB)                            \
  X(qDeriv, s.nD, 0, MJH_T_GLB, MJH_T_GLB)                                        \
  X(qLU, s.nD, 0, MJH_T_GLB, MJH_T_GLB)                                           \
  X(Dcdofdot, 6 * s.nD, 0, MJH_T_GLB, MJH_T_GLB)                                  \
  X(Dcvel, 6 * s.nB, 0, MJH_T_GLB, MJH_T_GLB)                                     \
  X(Dcacc, 6 * s.nB, 0, MJH_T_GLB, MJH_T_GLB)                                     \
  X(Dcfrcbody, 6 * s.nB, 0, MJH_T_GLB, MJH_T_GLB)                                 \
  X(rk_dX, 2 * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                                     \
  X(rk_act, 9 * s.na, 0, MJH_T_GLB, MJH_T_GLB)                                    \
  X(scratch, 8 * s.nefcmax + 8 * s.nv + 64, 0, MJH_T_GLB, MJH_T_GLB)              \
  /* per-stage time accumulators in microseconds (builds with -DMJH_PROFILE only) */ \
  X(energy, 2, 0, MJH_T_GLB, MJH_T_GLB)                                            \
  X(prof, 64, 0, MJH_T_GLB, MJH_T_GLB)

#define MJH_BATCH_INT_FIELDS(X)                                                   \
  /* counts: ncon, nefc, ne, nf, nl, solver_niter, nisland, paired, nJ, - */       \
  X(counts, 12, 12, MJH_T_COLLISION, MJH_T_END)                                   \
  /* per mjtWarning counter (include/mujoco/mjdata.h:74) */                       \
  X(warning, 8, 8, MJH_T_BEGIN, MJH_T_END)                                        \
  /* con_pair: index into the static pair list */                                 \
  X(con_pair, s.nconmax, MJH_LDS_CON, MJH_T_COLLISION, MJH_T_MAKE)                \
  X(con_geom, 2 * s.nconmax, 2 * MJH_LDS_CON, MJH_T_COLLISION, MJH_T_MAKE)        \
  X(con_dim, s.nconmax, MJH_LDS_CON, MJH_T_COLLISION, MJH_T_MAKE)                 \
  X(con_exclude, s.nconmax, MJH_LDS_CON, MJH_T_COLLISION, MJH_T_MAKE)             \
  X(con_efcadr, s.nconmax, MJH_LDS_CON, MJH_T_COLLISION, MJH_T_MAKE)              \
  /* mjContact flex[1] / elem[1] / vert[1] of every contact (models with flexes), -1 -1 -1 for geom : geom; candidate \
     table of a body : flex job (geom, vertex or element, parameter record, kind, selected flag; then the surviving leaves) */ \
  X(con_flex, MJH_CONFLEX * s.nconflex, 0, MJH_T_GLB, MJH_T_GLB)                            \
  X(flexcand_i, 6 * s.nflexcand, 0, MJH_T_GLB, MJH_T_GLB)                         \
  X(con_nodeb, (s.nconside ? (s.nconside + 1) * s.nconmax : 0), 0, MJH_T_GLB, MJH_T_GLB)       \
  X(moment_rownnz, s.nu, s.nu, MJH_T_TRANSMISSION, MJH_T_ACTUATION)               \
  X(moment_colind, s.nmoment, s.nmoment, MJH_T_TRANSMISSION, MJH_T_ACTUATION)     \
  X(efc_type, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                 \
  X(efc_id, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                   \
  X(efc_state, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                                \
  /* constraint islands (engine_island.c): island of every row; union-find work space */ \
  X(efc_island, s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                               \
  X(island_work, s.nefcmax + 2 * s.ntree + 8, 0, MJH_T_GLB, MJH_T_GLB)            \
  X(iscratch, 4 * s.nefcmax + 4 * s.nconmax + 64, 0, MJH_T_GLB, MJH_T_GLB)         \
  /* sparse constraint path: row pattern of every constraint (128-bit dof set, 4 words), CSR row addresses, \
     transpose (row addresses per dof, constraint index per entry), pattern of every row of the factor */ \
  X(sp_rowmask, 4 * s.sparse * s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                \
  X(sp_rowadr, (s.sparse + s.csr) * (s.nefcmax + 1), 0, MJH_T_GLB, MJH_T_GLB)               \
  X(sp_JTadr, (s.sparse + s.csr) * (s.nv + 1), 0, MJH_T_GLB, MJH_T_GLB)                     \
  X(sp_JTrow, s.nJmax, 0, MJH_T_GLB, MJH_T_GLB)                                   \
  /* explicit column indices of the compressed rows, dofs of the island being solved (mjh_csr.h) */ \
  X(sp_colind, s.csr * s.nJmax, 0, MJH_T_GLB, MJH_T_GLB)                          \
  X(csr_idof, s.csr * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                              \
  /* Newton on the explicit-index rows: elimination-tree parents | visit flags | lengths of the visiting lists | seeds, the \
     visiting lists (row r at (nv-1-r)(nv-2-r)/2), structural patterns of H and of the factor (xnw words per row) */ \
  X(xn_iw, 4 * s.xn * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                              \
  X(xn_LT, s.xn * s.xncap, 0, MJH_T_GLB, MJH_T_GLB)                               \
  X(xn_bits, 2 * s.xn * s.nv * s.xnw, 0, MJH_T_GLB, MJH_T_GLB)                    \
  X(sp_Lmask, 4 * s.sparse * s.nv, 0, MJH_T_GLB, MJH_T_GLB)                       \
  X(sp_Ladr, s.sparse * (s.nv + 1), 0, MJH_T_GLB, MJH_T_GLB)                      \
  /* structural pattern of every row of efc_AR (bit j of row i: the rows' Y patterns share a dof), 2 ints per word */ \
  X(sp_ARmask, 2 * s.nARw * s.nefcmax, 0, MJH_T_GLB, MJH_T_GLB)                   \
  /* 1: the env takes this step (no warning raised so far) -- written by the first kernel of a step */ \
  /* mjData.eq_active (user-switchable), first efc row of every equality this step */ \
  X(eq_active, s.neq, 0, MJH_T_GLB, MJH_T_GLB)                                    \
  X(eq_efcadr, s.neq, 0, MJH_T_GLB, MJH_T_GLB)                                    \
  X(active, 1, 0, MJH_T_GLB, MJH_T_GLB)                                           \
  /* launch balancing of the rollout kernels: cost = work estimate of env e over its last launch   \
     (sum over steps of 64 + nefc*(solver iterations + 4)); perm = launch order of the next launch, \
     most expensive first (workgroup w steps envs perm[w*nsub .. w*nsub+nsub)); wall = wall-clock \
     ticks (100 MHz >> 4) of the env's wavefront in its last launch (tail statistics) */          \
  X(cost, 1, 0, MJH_T_GLB, MJH_T_GLB)                                             \
  X(perm, 1, 0, MJH_T_GLB, MJH_T_GLB)                                             \
  X(wall, 1, 0, MJH_T_GLB, MJH_T_GLB)                                             \
  /* issue priority of the rollout wavefronts (mjh_step.h: rollout_env): environment 0's two slots hold the mean cost per  \
     environment of the previous launch (written by mjh_k_balance) and that launch's step count */ \
  X(prio_ref, 2, 0, MJH_T_GLB, MJH_T_GLB)

// indices into DBatch::counts
#define MJH_C_NCON 0
#define MJH_C_NEFC 1
#define MJH_C_NE 2
#define MJH_C_NF 3
#define MJH_C_NL 4
#define MJH_C_NITER 5
#define MJH_C_NISLAND 6
#define MJH_C_PAIRED 7     // stage_finish already produced mj_Euler's damped acceleration (qe) for this step
#define MJH_C_NJ 8         // non-zeros of the sparse constraint Jacobian (mjData.nJ)

// name : global home, n_name : per-env element count of the home, l_name : byte offset inside the
// workgroup's LDS block or -1, io_name : bit 0 = copy home -> LDS at kernel entry (live-in),
// bit 1 = copy LDS -> home at kernel exit (live-out)
struct DBatch {
  int nenv;
  int lds_bytes;     // LDS bytes per one-wavefront workgroup (0: no LDS plan, everything global)
  int dyn_off;       // [dyn_off, lds_bytes): free during MJH_T_MAKE..MJH_T_CONSTRAINT -> constraint arrays
  int dyn2_off;      // [dyn2_off, dyn_off): holds fields that die with MJH_T_MAKE; free from MJH_T_PROJECT on
  int nconlds;       // contact slots resident in LDS
  int soa;           // 0: fields are [nenv][count]; else = nenvpad, fields are [count][nenvpad]
  int mfma;          // 1: AR = Y Y' on the matrix cores (v_mfma_f64_16x16x4_f64): tolerance parity, not bit parity
  int pgs_mode;      // 0: the reference's PGS sweep bit for bit; 1: residual-update sweep (solve_pgs_resid: tolerance parity, opt-in)
  int xfrc_on;       // 1: xfrc_applied may be non-zero (mj_xfrcAccumulate runs; xipos stays readable at MJH_T_ACCEL)
  void* ccd_ws;      // convex narrowphase: [nenv][ccd_env_bytes] pair lists, contact records, overflow pages; null without convex pairs (mjh_convex.h)
#define X(name, cnt, lcnt, t0, t1) real* name; int n_##name; int l_##name; int io_##name;
  MJH_BATCH_REAL_FIELDS(X)
#undef X
#define X(name, cnt, lcnt, t0, t1) int* name; int n_##name; int l_##name; int io_##name;
  MJH_BATCH_INT_FIELDS(X)
#undef X
};

// argument block of the explicit-index CG solver's passes over the dofs (mjh_csrpass.h): plain pointers to unit-stride
// slices, so that wave 0 of a multi-wavefront workgroup can hand it to the helper wavefronts through LDS
enum { CSR_OP_WARM = 1, CSR_OP_START = 2, CSR_OP_GRAD = 3, CSR_OP_STEP = 4, CSR_OP_DIR = 5 };
struct CsrPass {
  int op, nv, flag;
  int lane0, width;            // the lanes that run the pass: wv_lane() in [lane0, lane0 + width)
  int pad_;
  real alpha;                  // step length (CSR_OP_STEP) or Hager-Zhang's beta (CSR_OP_DIR)
  real* grad; real* search;    // read by nothing but these passes, kept next to the products
  real* prod;                  // six product vectors, nv each: the addends of the iteration's ordered sums (LDS by plan)
  real* stage;                 // addends of the set-up's running sums (the sixth product vector)
  real* Ma; real* Mv; real* Mgrad;          // touched by these passes only: global memory, coalesced
  real* qacc; real* qfc;
  const real* qfs; const real* dinv; const real* Ms; const real* qws; const real* qas;
  const real* spJT; const real* force;
  const int* JTadr; const int* JTrow; const int* tree_island;
};

// argument block of the compressed rows' value pass (csr_row_values, mjh_csrpass.h)
struct CsrRowArgs { int nitems, ispyramid; SP<int> colind; SP<real> val; SP<const int> rowadr; SP<const int> items; };

// how the stage functions receive the two descriptors: read-only, constant address space
typedef const MJH_CONST_AS DModel& MREF;
typedef const MJH_CONST_AS DBatch& BREF;

// the workgroup's LDS block
#ifdef MJH_HOSTSIM
#define MJH_LDS_MAX (160 * 1024)
// (the emulated block ENDS at an inaccessible page: an access beyond the launch's allocation faults, as a flat access
// beyond a workgroup's LDS allocation does on the device -- tests/hostsim/hostsim.cpp: lds_block)
namespace mjhsim { extern thread_local char* g_lds; }
MJH_DEV char* mjh_lds() { return mjhsim::g_lds; }
#else
// The block is the launch's dynamic LDS allocation; the kernels declare no static __shared__
// data, so it starts at LDS offset 0 and its generic (flat) address is the shared aperture base
// itself.  Fields are reached through pointers that are LDS or global by plan, hence flat
// loads/stores throughout; reading the aperture register directly also sidesteps ROCm 7.2's
// mis-selection of the local->flat cast in divergent code ("V_CMP_NE_U32 0, src_shared_base").
MJH_DEV char* mjh_lds() {
  unsigned long long base;
  asm("s_mov_b64 %0, src_shared_base" : "=s"(base));
  return (char*)base;
}
#endif

// A field slice is handed out as a strided view (mjh_math.h: SP): stride 1 when the slice is LDS
// resident or the batch is laid out environment-major ([nenv][count]), stride nenvpad when the
// batch is laid out SoA across environments ([count][nenvpad], element i of env e at i*nenvpad+e).
// Local-address-space (ds_read/ds_write) access to a slice that the plan placed in LDS.  LDS
// returns are in order, so ds accesses pipeline (lgkmcnt(k)) where flat accesses must drain.
#ifdef MJH_HOSTSIM
template <class T> struct LP {
  T* p;
  template <class I> MJH_MEM T& operator[](I i) const { return p[i]; }
};
template <class T> MJH_DEV LP<T> mjh_local(T* flat) { return LP<T>{flat}; }
MJH_DEV long long mjh_lds_offset(const void* flat) { return (const char*)flat - mjh_lds(); }
#else
template <class T> struct LP {
  __attribute__((address_space(3))) T* p;
  template <class I> MJH_MEM __attribute__((address_space(3))) T& operator[](I i) const { return p[i]; }
};
// the workgroup's block starts at LDS address 0, so the local address is the offset from its base
template <class T> MJH_DEV LP<T> mjh_local(T* flat) {
  return LP<T>{(__attribute__((address_space(3))) T*)(unsigned)(size_t)((const char*)flat - mjh_lds())};
}
MJH_DEV long long mjh_lds_offset(const void* flat) { return (const char*)flat - mjh_lds(); }
#endif
// 1 if the contiguous slice lives in this workgroup's LDS block (flat LDS addresses are the shared
// aperture base + offset; 160 KB is the whole LDS of a CU)
template <class T> MJH_DEV int mjh_in_lds(const SP<T>& v) {
#ifdef MJH_HOSTSIM
  return 0;      // the emulation exercises the generic-pointer instantiation
#else
  const long long off = mjh_lds_offset((const void*)v.p);
  return v.s == 1 && off >= 0 && off < 160*1024;
#endif
}

template <class T>
MJH_DEV SP<T> mjh_gp(T* g, int n, int soa, int e) {
  return soa ? SP<T>{g + e, soa} : SP<T>{g + (size_t)e * (size_t)n, 1};
}
template <class T>
MJH_DEV SP<T> mjh_fp(T* g, int n, int l, int soa, int e, char* lds) {
  if (l >= 0) return SP<T>{(T*)(lds + l), 1};
  return mjh_gp(g, n, soa, e);
}
// the LDS block of the environment the calling lane works for: a workgroup (one wavefront) that
// steps several environments (sub-wave modes of mjh_modes.h) owns wv_sub-many consecutive blocks
#define MJH_LDS(B) (mjh_lds() + wv_sub()*(B).lds_bytes)
// env e's slice of field f (LDS if the plan placed it there, else its global home)
#define MJH_F(B, f, e) mjh_fp((B).f, (B).n_##f, (B).l_##f, (B).soa, (e), MJH_LDS(B))
// global home of env e's slice, whatever the plan says
#define MJH_G(B, f, e) mjh_gp((B).f, (B).n_##f, (B).soa, (e))
// record k (of `stride` elements) of a per-contact field: the first nconlds slots may live in LDS
template <class T>
MJH_DEV SP<T> mjh_cp(T* g, int n, int l, int soa, int e, int stride, int k, int nconlds, char* lds) {
  if (l >= 0 && k < nconlds) return SP<T>{(T*)(lds + l) + stride*k, 1};
  return mjh_gp(g, n, soa, e) + stride*k;
}
#define MJH_CON(B, f, e, stride, k) mjh_cp((B).f, (B).n_##f, (B).l_##f, (B).soa, (e), (stride), (k), (B).nconlds, MJH_LDS(B))

// constraint / contact enums used on device (include/mujoco/mjtype.h)
enum {
  MJH_JNT_FREE = 0, MJH_JNT_BALL = 1, MJH_JNT_SLIDE = 2, MJH_JNT_HINGE = 3,
  MJH_GEOM_PLANE = 0, MJH_GEOM_HFIELD = 1, MJH_GEOM_SPHERE = 2, MJH_GEOM_CAPSULE = 3,
  MJH_GEOM_ELLIPSOID = 4, MJH_GEOM_CYLINDER = 5, MJH_GEOM_BOX = 6, MJH_GEOM_MESH = 7,
  MJH_CNSTR_EQUALITY = 0, MJH_CNSTR_FRICTION_DOF = 1, MJH_CNSTR_FRICTION_TENDON = 2,
  MJH_CNSTR_LIMIT_JOINT = 3, MJH_CNSTR_LIMIT_TENDON = 4, MJH_CNSTR_CONTACT_FRICTIONLESS = 5,
  MJH_CNSTR_CONTACT_PYRAMIDAL = 6, MJH_CNSTR_CONTACT_ELLIPTIC = 7,
  MJH_STATE_SATISFIED = 0, MJH_STATE_QUADRATIC = 1, MJH_STATE_LINEARNEG = 2,
  MJH_STATE_LINEARPOS = 3, MJH_STATE_CONE = 4,
  MJH_SAMEFRAME_NONE = 0, MJH_SAMEFRAME_BODY = 1, MJH_SAMEFRAME_INERTIA = 2,
  MJH_SAMEFRAME_BODYROT = 3, MJH_SAMEFRAME_INERTIAROT = 4,
  MJH_WARN_INERTIA = 0, MJH_WARN_CONTACTFULL = 1, MJH_WARN_CNSTRFULL = 2,
  MJH_WARN_BADQPOS = 3, MJH_WARN_BADQVEL = 4, MJH_WARN_BADQACC = 5, MJH_WARN_BADCTRL = 6,
  MJH_WARN_UNSUPPORTED = 7,   // mjhip-only: an env reached a feature the GPU path does not implement
  MJH_TRN_JOINT = 0, MJH_TRN_JOINTINPARENT = 1, MJH_TRN_SLIDERCRANK = 2, MJH_TRN_TENDON = 3, MJH_TRN_SITE = 4, MJH_TRN_BODY = 5,
  MJH_GAIN_FIXED = 0, MJH_GAIN_AFFINE = 1, MJH_GAIN_MUSCLE = 2,
  MJH_BIAS_NONE = 0, MJH_BIAS_AFFINE = 1, MJH_BIAS_MUSCLE = 2,
  // sensor kinds (host translation of mjtSensor) and frame-object kinds (mjtObj)
  MJH_SENS_JOINTPOS = 0, MJH_SENS_JOINTVEL, MJH_SENS_TENDONPOS, MJH_SENS_TENDONVEL, MJH_SENS_ACTUATORPOS,
  MJH_SENS_ACTUATORVEL, MJH_SENS_ACTUATORFRC, MJH_SENS_JOINTACTFRC, MJH_SENS_BALLQUAT, MJH_SENS_BALLANGVEL,
  MJH_SENS_JOINTLIMITPOS, MJH_SENS_JOINTLIMITVEL, MJH_SENS_JOINTLIMITFRC, MJH_SENS_TENDONLIMITPOS,
  MJH_SENS_TENDONLIMITVEL, MJH_SENS_TENDONLIMITFRC, MJH_SENS_FRAMEPOS, MJH_SENS_FRAMEQUAT, MJH_SENS_FRAMEXAXIS,
  MJH_SENS_FRAMEYAXIS, MJH_SENS_FRAMEZAXIS, MJH_SENS_FRAMELINVEL, MJH_SENS_FRAMEANGVEL, MJH_SENS_FRAMELINACC,
  MJH_SENS_FRAMEANGACC, MJH_SENS_SUBTREECOM, MJH_SENS_SUBTREELINVEL, MJH_SENS_SUBTREEANGMOM, MJH_SENS_CLOCK,
  MJH_SENS_VELOCIMETER, MJH_SENS_GYRO, MJH_SENS_ACCELEROMETER, MJH_SENS_FORCE, MJH_SENS_TORQUE,
  MJH_SENS_MAGNETOMETER, MJH_SENS_TOUCH, MJH_SENS_INSIDESITE, MJH_SENS_TENDONACTFRC, MJH_SENS_RANGEFINDER,
  MJH_SENS_CONTACT, MJH_SENS_CAMPROJECTION, MJH_SENS_E_POTENTIAL, MJH_SENS_E_KINETIC,
  MJH_SENS_GEOMDIST, MJH_SENS_GEOMNORMAL, MJH_SENS_GEOMFROMTO,
  MJH_OBJ_BODY = 0, MJH_OBJ_XBODY = 1, MJH_OBJ_GEOM = 2, MJH_OBJ_SITE = 3, MJH_OBJ_NONE = 4, MJH_OBJ_CAMERA = 5,
  MJH_DYN_NONE = 0, MJH_DYN_INTEGRATOR = 1, MJH_DYN_FILTER = 2, MJH_DYN_FILTEREXACT = 3, MJH_DYN_MUSCLE = 4,
  MJH_SOL_PGS = 0, MJH_SOL_CG = 1, MJH_SOL_NEWTON = 2,
  MJH_INT_EULER = 0, MJH_INT_RK4 = 1, MJH_INT_IMPLICIT = 2, MJH_INT_IMPLICITFAST = 3,
  // pair_func: which narrowphase routine a static pair uses
  MJH_COL_PLANE_SPHERE = 0, MJH_COL_PLANE_CAPSULE = 1, MJH_COL_SPHERE_SPHERE = 2,
  MJH_EQ_CONNECT = 0, MJH_EQ_WELD = 1, MJH_EQ_JOINT = 2, MJH_EQ_TENDON = 3, MJH_EQ_FLEX = 4, MJH_EQ_FLEXVERT = 5,
  MJH_COL_SPHERE_CAPSULE = 3, MJH_COL_CAPSULE_CAPSULE = 4, MJH_COL_PLANE_CYLINDER = 5,
  MJH_COL_PLANE_BOX = 7, MJH_COL_SPHERE_BOX = 8, MJH_COL_SPHERE_CYLINDER = 9, MJH_COL_BOX_BOX = 10, MJH_COL_CAPSULE_BOX = 11,
  MJH_COL_UNSUPPORTED = 6,   // pair without a GPU collider (hfield, sdf): raises MJH_WARN_UNSUPPORTED if it survives the filter
  MJH_COL_CONVEX = 12,       // mjc_Convex: GJK / EPA / multicontact, one pair per lane (mjh_convex.h)
  MJH_COL_PLANE_CONVEX = 13, // mjc_PlaneConvex: plane against ellipsoid / mesh
};

// stage bits for partial forward evaluation (tests and per-stage profiling)
enum {
  MJH_STAGE_KINEMATICS = 1<<0,   // kinematics, comPos, tendon
  MJH_STAGE_INERTIA    = 1<<1,   // crb, factorM
  MJH_STAGE_COLLISION  = 1<<2,
  MJH_STAGE_MAKE       = 1<<3,   // makeConstraint
  MJH_STAGE_PROJECT    = 1<<4,   // Y, AR
  MJH_STAGE_TRANSMISSION = 1<<5,
  MJH_STAGE_VELOCITY   = 1<<6,   // ten/act velocity, comVel, passive, reference, rne
  MJH_STAGE_ACTUATION  = 1<<7,   // actuation + acceleration
  MJH_STAGE_CONSTRAINT = 1<<8,   // fwdConstraint (solve)
  MJH_STAGE_REFERENCE  = 1<<11,  // mj_referenceConstraint (efc_vel, efc_aref)
  MJH_STAGE_SENSOR     = 1<<12,  // mj_sensorPos/Vel/Acc (not part of MJH_STAGE_ALL: RK4 sub-steps skip it)
  MJH_STAGE_SENSPV     = 1<<13,  // only the position- and velocity-stage sensors (mj_sensorPos + mj_sensorVel: tail of mj_step1)
  MJH_STAGE_SENSACC    = 1<<14,  // only the acceleration-stage sensors (mj_sensorAcc: inside mj_step2)
  MJH_STAGE_IFACTIVE   = 1<<22,  // pipeline flag: skip environments whose `active` flag is 0
  MJH_STAGE_CHECKPV    = 1<<23,  // mj_checkPos + mj_checkVel first (head of mj_step1)
  MJH_STAGE_CHECKACC   = 1<<24,  // mj_checkAcc after the solve (mj_step2)
  MJH_STAGE_INTEGRATE  = 1<<25,  // integrate with the model's integrator, RK4 -> Euler (tail of mj_step2)
  MJH_STAGE_NOPARK     = 1<<26,  // step kernels: skip the global copy of M when nothing in the step will read it
  MJH_STAGE_FINISH     = 1<<10,  // qacc = M^-1 qfrc_constraint + qacc_smooth (tail of fwdConstraint)
  MJH_STAGE_ALL        = ((1<<9) - 1) | (1<<10) | (1<<11),
  MJH_STAGE_EULER      = 1<<9,   // mj_Euler + mj_advance (not part of mj_forward; for per-stage runs)
  MJH_STAGE_LDS        = 1<<21,  // host flag of mjhip_batch_forward: use the LDS residency plan (+ write-back)
  MJH_STAGE_WRITEBACK  = 1<<20,  // debug: copy LDS-resident fields to their global homes after every stage
};


// stage sets of the three kernels of the per-step pipeline
#define MJH_STAGES_SMOOTH_MASK (MJH_STAGE_KINEMATICS | MJH_STAGE_INERTIA | MJH_STAGE_TRANSMISSION | MJH_STAGE_VELOCITY | MJH_STAGE_ACTUATION)
#define MJH_STAGES_CONSTRAINT_MASK (MJH_STAGE_COLLISION | MJH_STAGE_MAKE | MJH_STAGE_PROJECT | MJH_STAGE_REFERENCE | MJH_STAGE_CONSTRAINT)

// arguments of the rollout kernel (device pointers; layouts of python/mujoco/rollout.cc:51-69)
struct RolloutArgs {
  int nstep;
  int has_ctrl;            // control_spec contains mjSTATE_CTRL
  int has_qfrc;            // control_spec contains mjSTATE_QFRC_APPLIED
  int ncontrol;            // mj_stateSize(control_spec)
  int qfrc_off;            // offset of qfrc_applied inside one control vector
  int mpos_off, mquat_off; // offsets of mocap_pos / mocap_quat inside one control vector, -1: absent
  int xfrc_off, eq_off, ud_off;   // offsets of xfrc_applied / eq_active / userdata, -1: absent
  int init;                // 1: load state0/warmstart0, clear warnings (start of a rollout)
  int t0;                  // (per-step kernels) index of this step inside control/state
  const real* state0;      // [nenv][nstate]        or null
  const real* warmstart0;  // [nenv][nv]            or null -> zeros
  const real* control;     // [nenv][nstep][ncontrol] or null
  real* state;             // [nenv][nstep][nstate] or null
  real* sensordata;        // [nenv][nstep][nsensordata] or null
  int env_offset;          // first env of this launch inside state0/control/state
  int nlaunch;             // > 0: only environments [0, nlaunch) take part, in identity order (partial launch)
  int pitch;               // steps of one environment's row in control / state / sensordata (>= tbase + nstep)
  int tbase;               // this launch's first step inside those rows (the host-array path launches a rollout in chunks)
};

