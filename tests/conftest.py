import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session")
def clean_up_pack() -> bytes:
  from meltingpot_amd import engine
  return engine.load_pack("clean_up")


@pytest.fixture(scope="session")
def commons_pack() -> bytes:
  from meltingpot_amd import engine
  return engine.load_pack("commons_harvest__open")


@pytest.fixture(scope="session")
def commons_closed_pack() -> bytes:
  from meltingpot_amd import engine
  return engine.load_pack("commons_harvest__closed")


@pytest.fixture(scope="session")
def territory_pack() -> bytes:
  from meltingpot_amd import engine
  return engine.load_pack("territory__rooms")


@pytest.fixture(scope="session")
def territory_open_pack() -> bytes:
  from meltingpot_amd import engine
  return engine.load_pack("territory__open")


@pytest.fixture(scope="session")
def commons_partnership_pack() -> bytes:
  from meltingpot_amd import engine
  return engine.load_pack("commons_harvest__partnership")


@pytest.fixture(scope="session")
def coop_mining_pack() -> bytes:
  from meltingpot_amd import engine
  return engine.load_pack("coop_mining")


@pytest.fixture(scope="session")
def coins_pack() -> bytes:
  from meltingpot_amd import engine
  return engine.load_pack("coins")


@pytest.fixture(scope="session")
def territory_inside_out_pack() -> bytes:
  from meltingpot_amd import engine
  return engine.load_pack("territory__inside_out")
