'use strict';
// Loads the N-API shim (tendrils_amd/lib/tendrils_hip.node) that binds the C ABI of
// include/tendrils_hip.h.  No fallback: without the addon or a gfx950 device calls throw.
const path = require('path');

module.exports = require(path.join(__dirname, '..', 'lib', 'tendrils_hip.node'));
