"""Python mirror of include/cygym_spec.h (bit layouts, counter columns, RNG sites).

tests/test_host_cpu.py::test_spec_mirror_matches_header parses the header and checks
every constant here against it, so the two cannot drift apart.
"""

# per-device dynamic flags (CDSimulatorComponents.py:217-243 flattened)
F_COMP, F_OWNED, F_KNOWN, F_REACH = 0x01, 0x02, 0x04, 0x08
F_NYA, F_EVOACT, F_BUSYC, F_WLADV = 0x10, 0x20, 0x40, 0x80
S_VALID = 0x40
S_KEEP = F_COMP | F_KNOWN | F_REACH | F_NYA | F_WLADV

# per-device static byte
D_DC, D_SERVER = 0x01, 0x02

# per-env flag bits
E_HAS_CKPT, E_EVO_INIT, E_DET_TRAIN, E_DET_RANDOM = 0x01, 0x02, 0x04, 0x08
E_PREV_SET, E_TOPO_OVF, E_BUSY_SAT = 0x10, 0x20, 0x40
E_DET_PENDING, E_UNPINNED = 0x100, 0x200
E_NX_SHIFT = 16   # bits 16..31 of ienv[I_FLAGS]: live entries of the env's extra-edge list

# ienv columns
(I_STEP_NUM, I_DEF_STEP, I_ATT_STEP, I_WORK_DONE, I_CKPT_CNT, I_REVERT_CNT,
 I_SCAN_CNT, I_COMP_CNT, I_EDGES_BLOCKED, I_EDGES_ADDED, I_FLAGS, I_RNG_TICK,
 I_LOG_TOTAL, I_DISCOVERED, I_LAST_NCOMP, I_LAST_ATYPE, I_COUNT) = range(17)

# fenv columns
D_DEF_COST, D_CLEAN_COST, D_PREV_ATT_POT, D_COUNT = range(4)

MODE_DEFENDER, MODE_ATTACKER = 0, 1
MODE_PARTIAL = 0x100
MODE_BASELINE_SHIFT = 16   # bits 16..18 of the mode word: (env.base_line code + 1) of this env for this tick, 0 = the config's
LOG_RING = 32
SCAN_WINDOW = 30
SLOW_SCAN_WINDOW = 256
HIST_RING = 2048
TRAIN_WINDOW = 2000

# trained detector (IsolationForest) -- layout of one env's forest, see cygym_spec.h
FOREST_TREES, FOREST_NODES, FOREST_HDR = 2, 512, 8
FOREST_WORDS = FOREST_HDR + FOREST_TREES * FOREST_NODES
DET_APL_N = 257

(SITE_STALL_REVERT, SITE_STALL_CLEAN, SITE_STALL_PATCH, SITE_STALL_SCAN,
 SITE_STALL_ISOLATE, SITE_PICK_BLOCK, SITE_PICK_UNBLOCK, SITE_PROBE_SRC,
 SITE_ZERODAY, SITE_ARR_CLIENT, SITE_ARR_SERVER, SITE_ARR_TIME,
 SITE_EVO_POISSON, SITE_EVO_COIN, SITE_EVO_PICK_IN, SITE_EVO_PICK_ACT,
 SITE_EVO_ATT, SITE_EVO_PA, SITE_SHUFFLE, SITE_DET_COIN, SITE_LAZY, SITE_DET_FIT) = range(1, 23)
SITE_ACTGEN = 64
SITE_EPS_TYPE = 65
SITE_GROUP_PICK = 66
SITE_SAMPLE = 67

POISSON_TABLE = 16
TRI_TABLE = 8
MAX_EXPLOITS = 6
MAX_EVENTS = 16
