"""Column layout of the state / observation matrix and of per-side arrays - the contract shared by the device
kernels (csrc/step_kernel.hpp), the Python host layer and every consumer (reference: gym/index_names.py:1-7)."""
CASH_INDEX = 0
INVENTORY_INDEX = 1
TIME_INDEX = 2
ASSET_PRICE_INDEX = 3

BID_INDEX = 0
ASK_INDEX = 1
