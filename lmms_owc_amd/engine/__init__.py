"""Host side of the MI355X hot path: weight packing, position bookkeeping and the batched
generate loop that drives libowc_hip.so (mirror of the reference's src/engine + src/models hot loop)."""
