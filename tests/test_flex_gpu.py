"""Flexes on the device: the cases of tests/test_flex_hostsim.py through the HIP library (free motion against the reference
linked with the kernels' libm routines is not needed here: a flex model's kinematics only adds and multiplies)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
import test_flex_hostsim as fh

pytestmark = pytest.mark.gpu


def test_jelly_free_motion_bit_exact_on_gpu(rb, hip_lib):
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "jelly.mjb"))
    fh._free_motion(rb, hip_lib, m)


def test_shell_bending_on_gpu(rb, hip_lib, tmp_path):
    xml = tmp_path / "shell.xml"
    xml.write_text(fh.flex_xml("6 6 1", "0 0 1", flex_attr='dim="2"',
                               flex_body='<edge equality="false" damping="10"/><contact contype="0" conaffinity="0"/>'
                                         '<elasticity young="3e5" poisson="0" thickness="1e-2" damping="1e-3" elastic2d="both"/><pin id="0 5 30 35"/>'))
    fh._free_motion(rb, hip_lib, rb.MjModel.from_xml_path(str(xml)))


def test_flex_contact_lists_exact_on_gpu(rb, hip_lib, tmp_path):
    fh._contact_lists(rb, hip_lib, tmp_path)


def test_flex_on_floor_keeps_fifty_contacts_on_gpu(rb, hip_lib, tmp_path):
    xml = tmp_path / "floor.xml"
    xml.write_text(fh.flex_xml("9 9 2", "0 0 .035"))
    maxcon, kinds = fh._resync_steps(rb, hip_lib, rb.MjModel.from_xml_path(str(xml)), pre=30, nstep=40)
    assert maxcon == 50 and kinds == {"vert"}


def test_jelly_on_the_capsule_on_gpu(rb, hip_lib):
    m = rb.MjModel.from_binary_path(os.path.join(GOLDEN, "jelly.mjb"))
    maxcon, kinds = fh._resync_steps(rb, hip_lib, m, pre=385, nstep=30)
    assert maxcon >= 20 and "elem" in kinds


def test_flex_against_an_actuated_body_on_gpu(rb, hip_lib, tmp_path):
    fh._actuated_body(rb, hip_lib, tmp_path)


def test_flex_island_next_to_rigid_islands_on_gpu(rb, hip_lib, tmp_path):
    fh._multi_island(rb, hip_lib, tmp_path)

def test_flex_edge_equality_constraints_on_gpu(rb, hip_lib, tmp_path):
    """mjEQ_FLEX rows (one per non-rigid edge) in front of the contact rows: bit-exact incl. CG iteration counts"""
    maxcon, kinds = fh._edge_equality(rb, hip_lib, tmp_path, nstep=60)
    assert maxcon > 0
