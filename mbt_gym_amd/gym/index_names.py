"""Column layout of the state / observation matrix and of per-side arrays - the contract shared by the device
kernels (csrc/step_kernel.hpp), the Python host layer and every consumer (reference: gym/index_names.py:1-7).

A state row is [cash, inventory, time, asset price, then the columns of the stochastic processes]; per-side arrays
(arrivals, fills, depths) are [bid, ask]."""
_STATE_COLUMNS = ("cash", "inventory", "time", "asset_price")
CASH_INDEX, INVENTORY_INDEX, TIME_INDEX, ASSET_PRICE_INDEX = range(len(_STATE_COLUMNS))
BID_INDEX, ASK_INDEX = range(2)
